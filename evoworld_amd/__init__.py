"""evoworld_amd -- MI355X-native (gfx950) implementation of EvoWorld's per-clip inference hot path.

Call surface mirrors the reference (JiahaoPlus/EvoWorld):
  evoworld_amd.unet.UNetSpatioTemporalConditionModel      <- evoworld/trainer/unet_plucker.py:30
  evoworld_amd.pipeline.StableVideoDiffusionPipeline      <- evoworld/pipeline/pipeline_evoworld.py:197
  evoworld_amd.scheduler.EulerDiscreteScheduler           <- diffusers EulerDiscreteScheduler (pipeline_evoworld.py:658,692,714)
  evoworld_amd.plucker.{equirectangular_to_ray, ray_c2w_to_plucker}  <- utils/plucker_embedding.py:56,221
  evoworld_amd.geometry.xyz_euler_to_{three_by_four,four_by_four}_matrix_batch <- dataset/CameraTrajDataset.py:643, utils/geometry.py:5
  evoworld_amd.reprojection.predictions_to_target_view    <- evoworld/reprojection/reproject_vggt_open3d_utils.py:1216
Compute runs in hand-written HIP kernels behind the C ABI of include/evoworld_hip.h
(evoworld_amd/libevoworld_hip.so).  There is NO CPU fallback: ops raise if the library is missing.
"""
__version__ = "0.1.0"
