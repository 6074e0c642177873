"""EulerDiscreteScheduler with the SVD-Xtend configuration (host-side schedule; the per-step arithmetic runs in
ew_euler_cfg_step).  Duck type used by the reference pipeline (evoworld/pipeline/pipeline_evoworld.py:658,692,714,433):
.set_timesteps(n, device=) .timesteps .sigmas .init_noise_sigma .order .scale_model_input(x,t) .step(eps,t,x).prev_sample

Restates diffusers==0.31.0 EulerDiscreteScheduler for the config shipped with stable-video-diffusion-img2vid-xt:
prediction_type=v_prediction, use_karras_sigmas, sigma_min=0.002, sigma_max=700, timestep_type=continuous,
timestep_spacing=leading, final_sigmas_type=zero.  The v-prediction form is pinned by the
reference's own training loss (c_skip/c_out at evoworld/trainer/train_evoworld.py:698-700)."""
from types import SimpleNamespace

import numpy as np
import torch


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, sigma_min=0.002, sigma_max=700.0, rho=7.0, prediction_type="v_prediction",
                 timestep_spacing="leading"):
        if prediction_type != "v_prediction":
            raise ValueError("only v_prediction (the SVD configuration) is implemented")
        self.config = SimpleNamespace(sigma_min=sigma_min, sigma_max=sigma_max, rho=rho, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, use_karras_sigmas=True,
                                      timestep_type="continuous", final_sigmas_type="zero")
        self.sigmas = None
        self.timesteps = None
        self._step_index = None
        self.num_inference_steps = None

    @classmethod
    def from_pretrained(cls, *_a, **_k):
        return cls()

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **_k):
        """num_inference_steps -> Karras schedule + final 0; or `sigmas` = a custom list INCLUDING its final value (the
        diffusers 0.31 custom-sigma branch that `retrieve_timesteps` reaches, pipeline_evoworld.py:180-190): used as given,
        timesteps = 0.25 * ln(sigma) over sigmas[:-1] (continuous timesteps + v-prediction)."""
        c = self.config
        if timesteps is not None:
            raise ValueError("custom `timesteps` are not supported by the continuous-timestep SVD configuration; pass `sigmas`")
        if sigmas is not None:
            sig = np.asarray(sigmas, dtype=np.float32)
            if sig.ndim != 1 or sig.size < 2 or not (sig[:-1] > 0).all():
                raise ValueError("`sigmas` must be a 1-D list of positive values followed by the final sigma")
            num_inference_steps = int(sig.size - 1)
        else:
            ramp = np.linspace(0, 1, num_inference_steps)
            min_inv, max_inv = c.sigma_min ** (1 / c.rho), c.sigma_max ** (1 / c.rho)
            sig = (max_inv + ramp * (min_inv - max_inv)) ** c.rho                     # Karras et al. (2022) eq. 5
            sig = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)                                       # kept on host: step scalars
        self.timesteps = torch.tensor([0.25 * float(np.log(s)) for s in sig[:-1]], dtype=torch.float32)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        self._step_index = None
        self.num_inference_steps = num_inference_steps

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return m
        return (m ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def _init_step(self):
        if self._step_index is None:
            self._step_index = 0

    def scale_model_input(self, sample, timestep=None):
        self._init_step()
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, **_k):
        """Host/torch form of the step (API parity; the fused HIP path is ew_euler_cfg_step)."""
        self._init_step()
        sigma = self.sigmas[self._step_index].to(sample.device)
        sigma_next = self.sigmas[self._step_index + 1].to(sample.device)
        sample = sample.to(torch.float32)
        x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        d = (sample - x0) / sigma
        prev = sample + d * (sigma_next - sigma)
        self._step_index += 1
        return SimpleNamespace(prev_sample=prev.to(model_output.dtype), pred_original_sample=x0)
