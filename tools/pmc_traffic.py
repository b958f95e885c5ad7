"""HBM-side bytes per kernel launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db profiles/rNN_hbm_traffic_pmc.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KB; FETCH_SIZE reports
half the bytes of wide coalesced reads and is doubled; Infinity-Cache hits are included."""
import json
import sqlite3
import sys

# gemm_pp: the ping-pong kernels = qkv / proj / fc1 / fc2 of every block and the patch embedding (49 of the encoder's 51
# GEMM launches, 99 % of its GEMM FLOPs); gemm_other: the neck's two GEMMs and the mask decoder's image-side GEMMs
GROUPS = [("gemm_pp", ("gemm_pp_kernel", "gemm_pp128_kernel")), ("gemm_other", ("gemm_f16_kernel", "gemm16_f16_kernel")),
          ("attn_window", ("attention_window_kernel",)), ("attn_global", ("attention_global",)), ("layernorm", ("layernorm",)),
          ("pre", ("preprocess_kernel",)), ("post", ("postprocess_kernel",))]


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return "other"


def short(name):
    """kernel name with its template arguments, whichever way the profiler spells it: demangled
    ('void dlimg::(anonymous namespace)::gemm_pp_kernel<0, 2>(dlimg::k::GemmArgs)') or mangled ('..14gemm_pp_kernelILi0ELi2EEEv..')"""
    import re
    m = re.search(r"(\w+_kernel)\s*(<[^>]*>)?", name.replace("(anonymous namespace)::", ""))
    if m and not m.group(1).startswith("_Z"):
        return m.group(1) + (m.group(2) or "").replace(" ", "")
    m = re.search(r"N_1\d+([a-z]\w*?_kernel)(I(?:Li\d+E)+E)?", name) or re.search(r"\d+([a-z]\w*?_kernel)(I(?:Li\d+E)+E)?", name)
    if m:
        args = re.findall(r"Li(\d+)E", m.group(2) or "")
        return m.group(1) + ("<" + ",".join(args) + ">" if args else "")
    return name[:48]


def collect(db_path, counter, by_grid=None):
    db = sqlite3.connect(db_path)
    out = {}
    for name, value, grid, wg in db.execute("select kernel_name, value, grid_size, workgroup_size from counters_collection "
                                            "where counter_name = ?", (counter,)):
        g = out.setdefault(group_of(name), [0, 0.0])
        g[0] += 1
        g[1] += value
        if by_grid is not None:
            e = by_grid.setdefault(f"{short(name)} wgs={int(grid // max(wg, 1))}", [0, 0.0])
            e[0] += 1
            e[1] += value
    return out


def stream_writer_shapes(db_path, counter, depth):
    """The encoder's stream writers -- patch embedding, proj, fc2 -- are ONE kernel (gemm_pp_kernel<0,2>) at ONE grid, so no
    by-name table can tell them apart; what can is their order on a stream: every pass launches patch, then (proj, fc2) per
    block, i.e. 2 * depth + 1 launches with that grid, always in this order.  Returns {shape: [launches, counter sum, grid]}
    for the grid with the most launches, or {} when a stream's launch count is not a whole number of passes (another kernel
    mix: nothing is guessed)."""
    db = sqlite3.connect(db_path)
    try:
        rows = db.execute("select c.kernel_name, c.value, c.grid_size, c.workgroup_size, c.dispatch_id, k.stream_id "
                          "from counters_collection c join rocpd_kernel_dispatch k on k.dispatch_id = c.dispatch_id and k.guid = c.guid "
                          "where c.counter_name = ?", (counter,)).fetchall()
    except sqlite3.Error:
        return {}
    seqs = {}
    for name, value, grid, wg, dispatch, stream in rows:
        if short(name) != "gemm_pp_kernel<0,2>":
            continue
        seqs.setdefault((int(grid // max(wg, 1)), stream), []).append((dispatch, value))
    if not seqs:
        return {}
    per_grid = {}
    for (wgs, _), seq in seqs.items():
        per_grid[wgs] = per_grid.get(wgs, 0) + len(seq)
    wgs = max(per_grid, key=per_grid.get)
    period = 2 * depth + 1
    out = {"gemm_patch": [0, 0.0, wgs], "gemm_proj": [0, 0.0, wgs], "gemm_fc2": [0, 0.0, wgs]}
    for (g, _), seq in seqs.items():
        if g != wgs:
            continue
        if len(seq) % period:
            return {}
        for i, (_, value) in enumerate(sorted(seq)):
            key = "gemm_patch" if i % period == 0 else ("gemm_proj" if (i % period) % 2 == 1 else "gemm_fc2")
            out[key][0] += 1
            out[key][1] += value
    return out


def main():
    fetch_grid, write_grid = {}, {}
    fetch, write = collect(sys.argv[1], "FETCH_SIZE", fetch_grid), collect(sys.argv[2], "WRITE_SIZE", write_grid)
    depth = int(sys.argv[5]) if len(sys.argv) > 5 else 12
    shape_f, shape_w = stream_writer_shapes(sys.argv[1], "FETCH_SIZE", depth), stream_writer_shapes(sys.argv[2], "WRITE_SIZE", depth)
    per_kernel = {}
    for g in sorted(set(fetch) | set(write)):
        n_f, kb_f = fetch.get(g, [0, 0.0])
        n_w, kb_w = write.get(g, [0, 0.0])
        per_kernel[g] = {
            "launches_in_run": n_f,
            "fetch_size_kb_per_launch_raw": kb_f / max(n_f, 1),
            "fetch_bytes_per_launch_corrected_x2": 2.0 * 1024.0 * kb_f / max(n_f, 1),
            "write_bytes_per_launch": 1024.0 * kb_w / max(n_w, 1),
        }
    doc = {
        "command": (sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 5 --warmup 1 --no-cpu-baseline")
                   + " (separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE)",
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); "
                "counters are in KB; includes Infinity-Cache hits",
        "per_kernel": per_kernel,
    }
    # the same per (kernel with its template arguments, workgroups of the launch): launches of different batch sizes and
    # shapes are different rows, so a row's bytes can be set against the algorithmic bytes of exactly that launch
    doc["by_kernel_and_grid"] = {
        k: {"launches_in_run": fetch_grid.get(k, [0, 0.0])[0],
            "fetch_bytes_per_launch_corrected_x2": 2.0 * 1024.0 * fetch_grid.get(k, [0, 0.0])[1] / max(fetch_grid.get(k, [0, 0.0])[0], 1),
            "write_bytes_per_launch": 1024.0 * write_grid.get(k, [0, 0.0])[1] / max(write_grid.get(k, [0, 0.0])[0], 1)}
        for k in sorted(set(fetch_grid) | set(write_grid))}
    # r06: the stream writers once more by shape (launch order on their stream; stream_writer_shapes)
    if shape_f and shape_w and set(shape_f) == set(shape_w):
        doc["by_shape"] = {
            k: {"kernel": "gemm_pp_kernel<0,2>", "grid": shape_f[k][2], "launches_in_run": shape_f[k][0],
                "fetch_bytes_per_launch_corrected_x2": 2.0 * 1024.0 * shape_f[k][1] / max(shape_f[k][0], 1),
                "write_bytes_per_launch": 1024.0 * shape_w[k][1] / max(shape_w[k][0], 1)}
            for k in shape_f}
        doc["by_shape_note"] = ("patch / proj / fc2 share gemm_pp_kernel<0,2> and its grid; told apart by their position in the "
                                f"{2 * depth + 1} launches of that kernel per pass on its stream (patch, then proj and fc2 per block)")
    doc.update(per_kernel.get("gemm_pp", {}))
    with open(sys.argv[3], "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in doc.get("by_shape", {}).items():
        print(f"{k:12s} launches {v['launches_in_run']:5d}  fetch {v['fetch_bytes_per_launch_corrected_x2'] / 1e6:8.2f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:8.2f} MB per launch (grid {v['grid']})")
    for g, v in per_kernel.items():
        print(f"{g:12s} launches {v['launches_in_run']:5d}  fetch {v['fetch_bytes_per_launch_corrected_x2'] / 1e6:8.2f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:8.2f} MB per launch")


if __name__ == "__main__":
    main()
