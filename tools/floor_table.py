#!/usr/bin/env python3
"""Per-launch practical floors of one U-Net forward, from a by-shape breakdown (EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 python bench.py 2> file).

For every launch:  t_mfma = algorithmic flops / R_MFMA,  t_hbm = ideal HBM bytes / R_HBM,  floor = max of the two, where
  R_MFMA = 1300 TFLOP/s   the rate the vendor's assembly GEMMs (hipBLASLt, rocprofv3-timed on the same boxes, DESIGN section 3.1 /
                          profiles/r04_a_hipblaslt_solutions.txt) reach on large fp16 problems with random operands: the practical
                          MFMA ceiling of this chip under its power-limited clock (0.52 of the 2.5 PFLOP/s headline peak);
  R_ATTN = 1225 TFLOP/s   spatial attention at head_dim 64: 0.49 of peak is the bound with PERFECT MFMA / softmax-VALU overlap
                          (1050 VALU vs 512 MFMA clocks per tile);
  R_HBM  = 5.3 TB/s       what the streaming read+write kernels of this build sustain (LayerNorm, GroupNorm apply); reads alone reach
                          6-7 TB/s in micro-benchmarks (profiles/r04_a_mb_feed_chunk_major.txt).
Ideal bytes: every operand once -- A (M x C_in x 2), W (N x K x 2), output (2 B, or 3 B for the split stream), residual operands
(3 B each), GEGLU output N/2 wide.

These three rates are YARDSTICKS OF THIS BUILD'S OWN CHOOSING (what the best kernels on this chip were measured at), not the hardware roofline.  Since
round 6 every row and the totals also carry the TRUE roofline the bench line is quoted against (MI355X_MICROARCH.md): 2500 TFLOP/s dense fp16
MFMA and 8 TB/s HBM -- column `peak floor` = max(flops / 2.5 PF, bytes / 8 TB/s) and `of peak` = peak floor / measured = the fraction of the
roofline the launch reaches -- and, with `--clock GHz` (the effective clock of the record run, `roofline.effective_clock_ghz`), the same peak
scaled by clock / 2.4.  `--vendor file` adds the rate torch / hipBLASLt reached on the same (M, N, K) where tools/experiments/exp30_vs_hipblaslt.py
measured it (bias-only fp16 GEMM, random operands, same box).
usage: python tools/floor_table.py breakdown.txt [--clock 2.0] [--vendor exp30.txt] [> table.md]"""
import re
import sys

R_MFMA, R_ATTN, R_HBM = 1300e12, 1225e12, 5.3e12
P_MFMA, P_HBM, NOMINAL_GHZ = 2500e12, 8.0e12, 2.4         # the true roofline (MI355X_MICROARCH.md)
TAPS = {"0": 1, "1": 9, "2": 3}


def gemm_bytes(M, N, K, mode, epi):
    cin = K // TAPS[mode]
    a = M * cin * 2
    w = N * K * 2
    split = bool(epi & 16)
    n_out = N // 2 if epi & 8 else N
    out = M * n_out * (3 if split else 2)
    res = 0
    if epi & 2 and not (mode != "0" and epi == 19):          # conv <.,19>: row-bias + split OUT, no residual operand
        res += M * N * (3 if split else 2)
    if epi & 4:
        res += M * N * (3 if split else 2)
    return a + w + out + res


def main(path, clock=None, vendor_path=None):
    vendor = {}
    if vendor_path:
        for l in open(vendor_path):
            m = re.match(r"M=\s*(\d+) N=\s*(\d+) K=\s*(\d+):.*?(\d+) TF/s\s+torch/hipBLASLt\s+[\d.]+ ms\s+(\d+) TF/s", l)
            if m:
                vendor[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = (int(m.group(4)), int(m.group(5)))
    rows = []
    for l in open(path):
        m = re.match(r"\s+(\S+?)(<[^>]*>)?\s+(.*?)\s*n=\s*(\d+) total\s+([\d.]+) ms\s+avg\s+([\d.]+) us(?:\s+([\d.]+) TF/s)?", l)
        if not m:
            continue
        name, tmpl, shape, n, tot = m.group(1), m.group(2) or "", m.group(3), int(m.group(4)), float(m.group(5))
        kv = dict(re.findall(r"(\w+)=(\d+)", shape))
        fl = by = 0.0
        rate = R_MFMA
        vk = None
        if name.startswith("conv_small_n"):
            M, N, K = int(kv["M"]), int(kv["N"]), int(kv["K"])
            fl, by = 2.0 * M * N * K, M * (K // 9 // 3 * 2) * 2 + N * K * 2 + M * N * 2       # the split rows [x_hi | x_lo] once (2/3 of the three K blocks)
        elif name.startswith("gemm"):
            e = [x.strip() for x in tmpl.strip("<>").split(",")]
            mode, epi = e[-2], int(e[-1])
            M, N, K = int(kv["M"]), int(kv["N"]), int(kv["K"])
            fl, by = 2.0 * M * N * K, gemm_bytes(M, N, K, mode, epi)
            vk = (M, N, K)
        elif name == "attn_spatial_kernel":
            S = int(kv["S"])
            n_seq, heads = 50, {9216: 5, 2304: 10, 576: 20, 144: 20}.get(S, 5)
            fl, by, rate = 4.0 * n_seq * heads * S * S * 64, n_seq * S * heads * 64 * 2 * 4, R_ATTN
        elif name == "ff320_kernel":
            M = 460800
            fl, by = 2.0 * M * 320 * 2560 + 2.0 * M * 1280 * 320, M * 320 * (2 + 3 + 3)
        elif name == "ln_kernel":
            by = int(kv["rows"]) * int(kv["C"]) * 5
        elif name == "gn_stats_kernel":
            by = int(kv["slabs"]) * int(kv["rows"]) * int(kv["C"]) * (3 if "+lo" in shape else 2)
        elif name == "gn_apply_kernel":
            by = int(kv["slabs"]) * int(kv["rows"]) * int(kv["C"]) * ((3 if "+lo" in shape else 2) + 2)
        elif name == "attn_temporal_kernel":
            by = 0           # per-level shapes not in the key: measured time kept as its own floor (HBM-bound at 5.2 TB/s)
        t_m, t_h = fl / rate * 1e3, by / R_HBM * 1e3
        floor = max(t_m, t_h) * n if (fl or by) else tot
        pfloor = max(fl / P_MFMA, by / P_HBM) * 1e3 * n if (fl or by) else 0.0
        rows.append((tot, name + tmpl, shape, n, floor, "mfma" if t_m >= t_h else "hbm", fl * n, by * n, pfloor, vendor.get(vk)))
    rows.sort(reverse=True)
    T, F, PF = sum(r[0] for r in rows), sum(r[4] for r in rows), sum(r[8] for r in rows)
    print(f"# floor table of one U-Net forward ({path}): measured {T:.1f} ms (sum of launches)")
    print(f"# against the TRUE roofline (2500 TFLOP/s dense fp16 MFMA, 8 TB/s HBM; per launch max(flops / peak, ideal bytes / peak)): {PF:.1f} ms = {PF / T:.3f} of the measured time"
          + (f"; at the effective clock of the record run ({clock:.2f} of {NOMINAL_GHZ} GHz): {PF * NOMINAL_GHZ / clock:.1f} ms = {PF * NOMINAL_GHZ / clock / T:.3f}" if clock else ""))
    print(f"# against this build's own yardsticks (MFMA {R_MFMA / 1e12:.0f} TF/s = the vendor-GEMM ceiling measured on this chip, attention {R_ATTN / 1e12:.0f} TF/s, HBM {R_HBM / 1e12:.1f} TB/s "
          f"= what the streaming kernels sustain): {F:.1f} ms = {F / T:.2f} of the measured time.  Ideal bytes = every operand once.")
    print("| kernel | shape | n | measured ms | TF/s | peak floor ms | of peak | yardstick floor ms | bound | measured / yardstick | vendor TF/s (this build, same call) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for tot, k, shape, n, floor, b, fl, by, pfloor, ven in rows:
        if tot < 0.25:
            continue
        print(f"| {k} | {shape} | {n} | {tot:.2f} | {(f'{fl / tot / 1e9:.0f}' if fl else '')} | {pfloor:.2f} | {pfloor / tot:.2f} | {floor:.2f} | {b} | {tot / max(floor, 1e-9):.2f} | "
              + (f"{ven[1]} ({ven[0]})" if ven else "") + " |")
    cls = {}
    for tot, k, shape, n, floor, b, fl, by, pfloor, ven in rows:
        c = ("spatial attention" if k.startswith("attn_spatial") else "fused FF (level 0)" if k.startswith("ff320") else
             "GroupNorm / LayerNorm / temporal attention" if k.split("<")[0] in ("ln_kernel", "gn_stats_kernel", "gn_apply_kernel", "gn_finalize_kernel", "attn_temporal_kernel")
             else "GEMM / conv, MFMA-bound at the floor" if b == "mfma" else "GEMM / conv, HBM-bound at the floor")
        a = cls.setdefault(c, [0.0, 0.0, 0.0])
        a[0] += tot
        a[1] += floor
        a[2] += pfloor
    print("\n| class | measured ms | peak floor ms | of peak | yardstick floor ms | gap to yardstick ms |\n|---|---|---|---|---|---|")
    for c, (a, b, pf) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
        print(f"| {c} | {a:.1f} | {pf:.1f} | {pf / a:.2f} | {b:.1f} | {a - b:.1f} |")
    print(f"| **total** | {T:.1f} | {PF:.1f} | {PF / T:.2f} | {F:.1f} | {T - F:.1f} |")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("breakdown")
    ap.add_argument("--clock", type=float, default=None, help="effective clock of the record run in GHz (roofline.effective_clock_ghz)")
    ap.add_argument("--vendor", default=None, help="output of tools/experiments/exp30_vs_hipblaslt.py on the same box")
    a = ap.parse_args()
    main(a.breakdown, a.clock, a.vendor)
