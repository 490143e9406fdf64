"""profiles/<round>_pmc_summary.json from rocprofv3 --pmc passes of bench.py (SQ/GRBM, FETCH_SIZE, WRITE_SIZE - separate
runs, as MI355X_MICROARCH.md prescribes).  The render rows come from passes WITHOUT the training section
(`--train-rays 0`), so that every dispatch of a render kernel in them is one of the bench's frame launches; the training
rows come from three more passes with it.

    python tools/make_pmc_profile.py <sq_dir> <fetch_dir> <write_dir> [<train_sq_dir> <train_fetch_dir> <train_write_dir>]
"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import summarise

KERNELS = {"fp32": "mlp_fwd_kernel<256, 8, false, false>", "f16x3": "mlp_fwd_bf16_kernel<256, 8, 2, false, 1>",
           "bf16x6": "mlp_fwd_bf16_kernel<256, 8, 3, false, 0>", "bf16x3": "mlp_fwd_bf16_kernel<256, 8, 2, false, 0>"}
TRAIN_KERNELS = {"fwd_train_f16x3": "mlp_fwd_bf16_kernel<256, 8, 2, true, 1>", "fwd_train_bf16x6": "mlp_fwd_bf16_kernel<256, 8, 3, true, 0>", "dgrad_f16x3": "mlp_bwd_bf16_kernel<256, 8, 2, false, 1>", "dgrad_bf16x6": "mlp_bwd_bf16_kernel<256, 8, 3, false, 0>",
                 "wgrad_f16x3_wide": "mlp_wgrad_f16_kernel", "wgrad_bf16x6_wide": "mlp_wgrad_bf16_kernel<3, 0>", "wgrad_fp32_wide": "mlp_wgrad_kernel(", "wgrad_f16x3_narrow": "mlp_wgrad_direct_f16_kernel", "wgrad_fp32_narrow": "mlp_wgrad_direct_kernel", "fwd_train_fp32": "mlp_fwd_kernel<256, 8, false, true>",
                 "dgrad_fp32": "mlp_bwd_kernel<256, 8, false"}
ALG_BYTES = {1048576: 1048576 * (12 + 16) + 16384 * 12, 3145728: 3145728 * (12 + 16) + 16384 * 12}  # x, raw, dirs


def main(sq, fetch, write, tsq=None, tfetch=None, twrite=None):
    out = {"command": "render rows: rocprofv3 --pmc <counters> --output-format csv -- python bench.py --steps 3 --warmup 1 "
                      "--cpu-rays 0 --train-rays 0 --no-pmc (three separate passes: SQ/GRBM, FETCH_SIZE, WRITE_SIZE; every "
                      "dispatch of a render kernel is a frame launch of the bench); training rows: the same three passes of "
                      "python bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 3 --no-pmc",
           "note": "FETCH_SIZE/WRITE_SIZE are KB; mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * "
                   "1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / 8 / duration.  MI355X_MICROARCH.md: FETCH_SIZE "
                   "under-reports wide 16 B/lane streams by 2x; these kernels read 4 B/lane inputs and take their weights "
                   "from L2 (the stream is re-read by every workgroup and stays on chip), so the figure is reported as is.",
           "kernels": {}}
    for prec, name in KERNELS.items():
        a, f, w = (summarise(name, [d]) for d in (sq, fetch, write))
        launches = {}
        tot_f = tot_w = tot_n = 0
        for key, e in a.items():
            n = int(key.split("grid=")[1]) // 512 * 128  # 512 threads per workgroup, 128 samples (non-persistent kernels)
            c = e["counters"]
            fk, wk = f[key]["counters"]["FETCH_SIZE"], w[key]["counters"]["WRITE_SIZE"]
            label = f"n={n}"
            if n not in ALG_BYTES:   # persistent kernels: the grid is one workgroup per CU whatever n is
                label = "coarse n=1048576 and fine n=3145728 launches averaged (persistent grid)"
            launches[label] = {"duration_ms": e["avg_seconds"] * 1e3, "effective_clock_GHz": e["effective_clock_ghz"],
                                  "mfma_pipe_busy_frac": e["mfma_busy_frac"],
                                  "wave_cycles_parked_frac": e.get("sq_wait_any_per_wave_cycle"),
                                  "wave_cycles_issue_stall_frac": e.get("sq_wait_inst_any_per_wave_cycle"),
                                  "wave_cycles_issuing_frac": e.get("sq_active_inst_any_per_wave_cycle"),
                                  "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                                  "GRBM_GUI_ACTIVE_sum_over_8_XCD": c["GRBM_GUI_ACTIVE"],
                                  "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "algorithmic_bytes": ALG_BYTES.get(n)}
            tot_f += fk * 1024
            tot_w += wk * 1024
            tot_n += 1
        if tot_n:
            out["kernels"][prec] = {"kernel": "snerf::" + name, "launches": launches,
                                    "avg_launch": {"fetch_bytes_as_reported": tot_f / tot_n, "write_bytes": tot_w / tot_n,
                                                   "hbm_bytes": (tot_f + tot_w) / tot_n,
                                                   "algorithmic_bytes": sum(ALG_BYTES.values()) / 2}}
    out["training_kernels"] = {}
    if tsq is None:
        print(json.dumps(out, indent=1))
        return
    for tag, name in TRAIN_KERNELS.items():
        a, f, w = (summarise(name, [d]) for d in (tsq, tfetch, twrite))
        for key, e in a.items():
            c = e["counters"]
            out["training_kernels"][f"{tag} grid={key.split('grid=')[1]}"] = {
                "kernel": "snerf::" + name, "duration_ms": e["avg_seconds"] * 1e3, "effective_clock_GHz": e["effective_clock_ghz"],
                "mfma_pipe_busy_frac": e["mfma_busy_frac"], "wave_cycles_parked_frac": e.get("sq_wait_any_per_wave_cycle"),
                "FETCH_SIZE_KB": f.get(key, {}).get("counters", {}).get("FETCH_SIZE"),
                "WRITE_SIZE_KB": w.get(key, {}).get("counters", {}).get("WRITE_SIZE")}
    print(json.dumps(out, indent=1))


SMPL_KERNELS = {"warp_fwd": "warp_fwd_resident_kernel<256, 16, false>", "warp_fwd_train": "warp_fwd_resident_kernel<256, 8, true>",
                "warp_bwd_dgrad": "warp_bwd_light_kernel", "render_fp32_per_sample_dirs": "mlp_fwd_kernel<256, 8, false, false>",
                "fwd_train_fp32": "mlp_fwd_kernel<256, 8, false, true>", "dgrad_fp32_input_grad": "mlp_bwd_kernel<256, 8, true",
                "wgrad_fp32_wide": "mlp_wgrad_kernel(", "wgrad_fp32_narrow": "mlp_wgrad_narrow_kernel",
                "wgrad_fp32_direct": "mlp_wgrad_direct_kernel"}


def generic(sq, fetch, write, kernels, command):
    """Per-kernel rows (every distinct grid separately) of one workload: duration, clock, MFMA-pipe busy, HBM bytes."""
    out = {"command": command, "note": "see the main summary's note for units and corrections", "kernels": {}}
    for tag, name in kernels.items():
        a, f, w = (summarise(name, [d]) for d in (sq, fetch, write))
        for key, e in a.items():
            out["kernels"][f"{tag} grid={key.split('grid=')[1]}"] = {
                "kernel": "snerf::" + name, "duration_ms": e["avg_seconds"] * 1e3, "effective_clock_GHz": e["effective_clock_ghz"],
                "mfma_pipe_busy_frac": e["mfma_busy_frac"], "wave_cycles_parked_frac": e.get("sq_wait_any_per_wave_cycle"),
                "FETCH_SIZE_KB": f.get(key, {}).get("counters", {}).get("FETCH_SIZE"),
                "WRITE_SIZE_KB": w.get(key, {}).get("counters", {}).get("WRITE_SIZE")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--smpl-nerf":
        generic(*sys.argv[2:5], SMPL_KERNELS,
                "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --workload smpl_nerf --steps 3 --warmup 1 "
                "--cpu-rays 0 --train-steps 3 --no-pmc --no-alt --points= (three separate passes: SQ/GRBM, FETCH_SIZE, WRITE_SIZE)")
    else:
        main(*sys.argv[1:7])
