"""Counter-derived MFMA utilisation per kernel group of one profiled bench run:

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES \\
        -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/m_results.db profiles/rNN_mfma_util_pmc.json

SQ_INSTS_VALU_MFMA_MOPS_F16 counts 512 FLOP per unit (checked: 4096x768x3072 GEMM = 19.33 GFLOP -> 37,748,736 units);
utilisation = units * 512 / (kernel duration * 2.5 PFLOP/s).  Kernels run one at a time under the profiler, so these
are per-kernel figures, not the overlapped whole-job rate."""
import json
import sqlite3
import sys

GROUPS = [("gemm_pp", ("gemm_pp_kernel", "gemm_pp128_kernel")), ("gemm_other", ("gemm_f16_kernel", "gemm16_f16_kernel")),
          ("attention_window", ("attention_window_kernel",)), ("attention_global", ("attention_global",))]
PEAK = 2.5e15


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return None


def main():
    db = sqlite3.connect(sys.argv[1])
    acc = {}
    q = "select kernel_name, dispatch_id, counter_name, value, (end - start) from counters_collection"
    seen = {}
    for name, disp, counter, value, dur in db.execute(q):
        g = group_of(name)
        if g is None:
            continue
        a = acc.setdefault(g, {"launches": 0, "ns": 0.0})
        if (g, disp) not in seen:
            seen[(g, disp)] = True
            a["launches"] += 1
            a["ns"] += dur
        a[counter] = a.get(counter, 0.0) + value
    out = {"command": (sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 5 --warmup 1 --no-cpu-baseline")
                      + " under rocprofv3 --pmc (kernels serialised)",
           "note": "SQ_INSTS_VALU_MFMA_MOPS_F16 = 512 FLOP per unit; peak 2.5 PFLOP/s dense f16", "per_kernel": {}}
    tot_flop = tot_ns = 0.0
    for g, a in acc.items():
        flop = a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512.0
        tot_flop += flop
        tot_ns += a["ns"]
        out["per_kernel"][g] = {
            "launches": a["launches"], "avg_us": a["ns"] / a["launches"] / 1e3,
            "mfma_tflops": flop / a["ns"] / 1e3, "mfma_util": flop / (a["ns"] * 1e-9) / PEAK,
            "mfma_busy_over_cu_busy": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(a.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0),
        }
    out["encoder_mfma_kernels"] = {"mfma_tflops": tot_flop / tot_ns / 1e3, "mfma_util": tot_flop / (tot_ns * 1e-9) / PEAK}
    with open(sys.argv[2], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
