#!/usr/bin/env python3
"""Per-kernel roofline table from the committed rocprofv3 artifacts (profiles/)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, L, HD, DM, D = 256, 120, 121, 1024, 512, 198
PEAK_TF, PEAK_I8, PEAK_HBM = 2500.0, 5000.0, 8.0  # TFLOP/s bf16 dense, TOP/s int8 dense, TB/s (MI355X_MICROARCH.md)

ALG = {  # kernel tag -> (what, algorithmic FLOPs per launch, algorithmic HBM bytes per launch)
    # precision 9: the layer's int8 input rows in (2 B x 512 per token), the attention output as int8 rows out (2 B x 1024), the three
    # projections' int8 weights once (2 B x 3 x 1024 x 512)
    "attn_layer_i8": ("Q/K/V projections + softmax + PV (one layer), int8 slices", B * (2 * L * DM * 3 * HD + 4 * L * L * HD),
                             2 * B * L * DM + 2 * B * L * HD + 2 * 3 * HD * DM),
    # precision 9: the attention output (2 B x 1024), the residual rows in and the layer's rows out (2 B x 512 each) per token; the
    # LayerNorm-1 rows and the hidden rows stay in LDS
    "tail_kernel<1,true,true,true,4,true>|layer_tail": ("fc+LN, FFN-1, FFN-2+LN (one layer; precision 9: all three contractions on int8 slices, FFN operands resident in LDS; int8 peak)",
                               B * L * (2 * HD * DM + 4 * DM * DM), 2 * B * L * (HD + DM + DM) + 2.1e6),
    # the split-bf16 operand (x | x_cond, 4 B per value) in, int8 rows (2 B per value) out
    "EpiEmbed": ("embed GEMM + time token + pos-emb (split-bf16 in, int8 rows out)", 2 * B * T * 2 * D * DM, 4 * B * T * 2 * D + 2 * B * L * DM),
    # int8 rows in; x read + written (fp32), the next step's embed operand written (split-bf16)
    "EpiOut": ("linear_out on int8 slices + DDPM posterior", 2 * B * T * DM * D, 2 * B * L * DM + 3 * 4 * B * T * D),
}
ATTN_CORE_FLOPS = B * 4 * L * L * HD  # QK^T + PV alone: north_star's "attention-GEMM"

ALG_P3 = {  # split-bf16 (precision 3: what `auto` runs on a checkpoint whose chain amplifies operand rounding): every row 4 B per value
    "qkv_attn_kernel": ("fused Q/K/V projections + attention (one layer), split-bf16; K and V^T cross L2, Q stays in registers", B * (2 * L * DM * 3 * HD + 4 * L * L * HD),
                        4 * B * L * DM + 4 * B * L * HD + 4 * 3 * HD * DM),
    "layer_tail_kernel": ("fc+LN, FFN-1, FFN-2+LN (one layer), split-bf16, 128-token eight-wave workgroups", B * L * (2 * HD * DM + 4 * DM * DM),
                          4 * B * L * (HD + DM + DM) + 4.2e6),
    "EpiEmbed": ("embed GEMM + time token + pos-emb (split-bf16 in and out)", 2 * B * T * 2 * D * DM, 4 * B * T * 2 * D + 4 * B * L * DM),
    "EpiOut": ("linear_out (split-bf16) + DDPM posterior", 2 * B * T * DM * D, 4 * B * L * DM + 3 * 4 * B * T * D),
}


def table(rnd, alg, stats_name, traffic_name, pmc_name, peak_of):
    stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", rnd + stats_name))))
    traffic = json.load(open(os.path.join(ROOT, "profiles", rnd + traffic_name)))["kernels"]
    pmc = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", rnd + pmc_name)))}
    rows = []
    for tag, (what, flops, abytes) in alg.items():
        tags = tag.split("|")
        st = next(r for r in stats if any(t in r["Name"].replace(" ", "") for t in tags))
        us = float(st["AverageNs"]) / 1e3
        tr = next((v for k, v in traffic.items() if any(t in k for t in tags)), None)
        pm = next((v for k, v in pmc.items() if any(t in k for t in tags)), {})
        hbm = tr["hbm_bytes_per_launch"] if tr else None
        peak = peak_of(tag, st["Name"])
        rows.append((tags[0], what, us, float(st["Percentage"]), flops / us / 1e6, flops / us / 1e6 / peak, 3 * flops / us / 1e6 / peak,
                     abytes / 1e6, (hbm or 0) / 1e6, (hbm or 0) / us / 1e6, pm.get("us_sq_pass", ""), pm.get("clock_ghz_grbm_pass", ""), pm.get("mfma_busy_frac_at_that_clock", "")))
    return rows


HEAD = ["| kernel | what | avg µs | % of GPU time | algorithmic TFLOP/s (TOP/s) | frac of the dtype's peak | MFMA-pipe frac (×3) | algorithmic MB | measured HBM MB | HBM TB/s | µs in the SQ PMC pass | clock in the GRBM pass (GHz) | MFMA-busy at that clock |",
        "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]


def fmt(r):
    return f"| `{r[0]}` | {r[1]} | {r[2]:.0f} | {r[3]:.1f} | {r[4]:.0f} | {r[5]:.3f} | {r[6]:.3f} | {r[7]:.0f} | {r[8]:.0f} | {r[9]:.2f} | {r[10]} | {r[11]} | {r[12]} |"


def main():
    """Everything below is recomputed from the files under profiles/ (kernel stats of the un-profiled-clock run, the PMC summary with
    its own per-pass durations, the traffic json): no number comes from prose."""
    rnd = os.environ.get("EGOEGO_ROUND", "r05")
    rows = table(rnd, ALG, "_bench_b256_t120_kernel_stats.csv", "_traffic.json", "_pmc_per_kernel.csv",
                 lambda tag, name: PEAK_I8 if ("i8" in tag or "tail_kernel" in tag or "gemm_i8" in name) else PEAK_TF)  # (precision 9: the tail and linear_out issue int8 MFMAs only)
    out = [f"# Per-kernel roofline, round {int(rnd[1:])} (B=256, T=120, precision 9: int8-slice attention layer, fc, FFN and linear_out; split-bf16 embed; from the files in this directory)", "",
           "MFMA bound: 2.5 PFLOP/s dense bf16, 5 POP/s dense int8; both split-bf16 and the int8 slices issue 3 MFMAs per algorithmic product, so the algorithmic fraction is capped at 33 %.",
           "HBM bound: 8 TB/s.  `traffic` = (2·FETCH_SIZE + WRITE_SIZE)·1024 from the PMC passes.  `avg µs` is the un-profiled rocprofv3 --stats run; the last three columns come from the",
           "PMC passes alone (their own durations: `*_pmc_per_kernel.csv`): clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock).", "",
           ] + HEAD
    out += [fmt(r) for r in rows]
    a = rows[0]
    out += ["", f"Attention core alone (QK^T + PV, what north_star words as the \"attention-GEMM roofline\"): {ATTN_CORE_FLOPS / 1e9:.2f} GOP of the attention-layer kernel's "
            f"{ALG['attn_layer_i8'][1] / 1e9:.1f} ({100 * ATTN_CORE_FLOPS / ALG['attn_layer_i8'][1]:.1f} %); it has no launch of its own (K, V, Q and the probabilities never leave the CU), "
            f"so its roofline fraction is the kernel's: {a[5]:.3f} of the int8 peak.  The per-phase trace (tools/attn_layer_trace.py) puts S^T + softmax + PV at 6-7 of a workgroup's 38-45 µs."]
    if os.path.exists(os.path.join(ROOT, "profiles", rnd + "_traffic_p3.json")):
        rows3 = table(rnd, ALG_P3, "_bench_b256_t120_p3_kernel_stats.csv", "_traffic_p3.json", "_pmc_per_kernel_p3.csv", lambda tag, name: PEAK_TF)
        step_us = sum(r[2] * (4 if r[0] in ("qkv_attn_kernel", "layer_tail_kernel") else 1) for r in rows3)
        out += ["", f"## Split-bf16 (precision 3) — what `auto` runs on a trained-like checkpoint (B=256, T=120; `{rnd}_bench_b256_t120_p3_kernel_stats.csv`, `{rnd}_pmc_per_kernel_p3.csv`, `{rnd}_traffic_p3.json`)", ""] + HEAD
        out += [fmt(r) for r in rows3]
        fl = sum(v[1] * (4 if k in ("qkv_attn_kernel", "layer_tail_kernel") else 1) for k, v in ALG_P3.items())
        out += ["", f"Step = embed + 4 x (attention layer + tail) + linear_out = {step_us:.0f} µs of kernel time; {fl / 1e9:.1f} algorithmic GFLOP per step -> {fl / step_us / 1e6:.0f} TFLOP/s = "
                f"{fl / step_us / 1e6 / PEAK_TF:.3f} of the bf16 peak ({3 * fl / step_us / 1e6 / PEAK_TF:.3f} of the matrix pipe at three MFMAs per product)."]
    text = "\n".join(out) + "\n"
    open(os.path.join(ROOT, "profiles", rnd + "_roofline.md"), "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
