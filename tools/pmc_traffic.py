"""Summarise the counter passes of tools/pmc_bench.sh.

    python tools/pmc_traffic.py gpurun_out/pmc_bench_<tag> profiles/r02

writes <prefix>_pmc_traffic.json (memory-side bytes per launch of the SpMM hop and of one fused
top-K call, read by bench.py) and <prefix>_pmc_sq_summary.txt (SQ busy / MFMA-busy / VALU-busy
per kernel).  FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B read
requests at 64 B, so reads of 16 B per lane are doubled (all reads of these kernels are)."""
import collections
import csv
import json
import sys

src, prefix = sys.argv[1], sys.argv[2]

SPMM = ("spmm_rows_kernel",)
TOPK = ("score_topk_kernel", "topk_merge_sorted_kernel", "topk_merge_kernel", "refine_kernel", "row_stats_kernel",
        "table_stats_kernel", "to_half_rows_kernel", "strata_max_kernel", "strata_threshold_kernel", "collect_kernel",
        "refine_wide_kernel", "mask_hist_kernel", "heavy_thresh_kernel", "topk_merge_heads_kernel")


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


line = json.loads(open(f"{src}/bench_line.json").read())
fetch, write = per_kernel(f"{src}/fetch.csv"), per_kernel(f"{src}/write.csv")
calls = line["pmc_pass_calls"]["score_topk_call"]
groups = {}
# SpMM hop: mean per dispatch of the row kernel plus its reduce (every dispatch in the run is a hop
# over the same graph at the same width)
g = {"kernels": {}, "fetch_size_kb": 0.0, "write_size_kb": 0.0}
for table, key in ((fetch, "fetch_size_kb"), (write, "write_size_kb")):
    for k, c in table.items():
        if any(s in k for s in SPMM):
            for cname, (v, n) in c.items():
                g[key] += v / n
                g["kernels"].setdefault(short(k), {})[key] = round(v / n, 1)
groups["spmm_hop"] = g
# one fused top-K call: everything its kernels fetched and wrote in the run, over the calls made
g = {"kernels": {}, "fetch_size_kb": 0.0, "write_size_kb": 0.0, "calls_in_run": calls}
for table, key in ((fetch, "fetch_size_kb"), (write, "write_size_kb")):
    for k, c in table.items():
        if any(s in k for s in TOPK):
            for cname, (v, n) in c.items():
                g[key] += v / calls
                d = g["kernels"].setdefault(short(k), {})
                d[key] = round(v / calls, 1)
                d["dispatches_per_call"] = round(n / calls, 2)
groups["score_topk_call"] = g
for g in groups.values():
    g["fetch_size_kb"], g["write_size_kb"] = round(g["fetch_size_kb"], 1), round(g["write_size_kb"], 1)
    g["traffic_bytes"] = round(1e3 * (2 * g["fetch_size_kb"] + g["write_size_kb"]))
    g["formula"] = "1000 * (2 * FETCH_SIZE + WRITE_SIZE)"
doc = {"shape": line["config"]["workload"].split(" ")[0], "emb_dim": 64 if "emb_dim 64" in line["config"]["workload"] else 128,
       "kernel_source_digest": line["kernel_source_digest"], "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 bench.py --pmc-pass",
       "groups": groups}
json.dump(doc, open(f"{prefix}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v["traffic_bytes"] for k, v in groups.items()}))

with open(f"{prefix}_pmc_sq_summary.txt", "w") as out:
    out.write("SQ counters per kernel, mean per dispatch (SQ_*_CYCLES of waves and instructions count quad-cycles,\n"
              "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles summed over the shader engines).\n"
              "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE-equivalent: SQ_BUSY_CYCLES / n_se); the ratios\n"
              "valu_active/wave = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES and wait/wave = SQ_WAIT_ANY / SQ_WAVE_CYCLES are unit-free.\n\n")
    for fname in ("sq_bench", "sq_proj"):
        t = per_kernel(f"{src}/{fname}.csv")
        out.write(f"== {fname} ==\n")
        for k, c in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
            if "at::native" in k or "rocclr" in k:
                continue
            m = {cn: v / n for cn, (v, n) in c.items()}
            n = next(iter(c.values()))[1]
            wc = max(m.get("SQ_WAVE_CYCLES", 0), 1)
            out.write(f"{short(k)[:60]:60s} x{n:<4d} busy {m.get('SQ_BUSY_CYCLES', 0):12.0f}  wave_cyc {wc:13.0f}  "
                      f"valu/wave {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.3f}  any/wave {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f}  "
                      f"wait/wave {m.get('SQ_WAIT_ANY', 0) / wc:5.3f}  stall/wave {m.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f}  "
                      f"mfma_busy_cyc {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):12.0f}  waves {m.get('SQ_WAVES', 0):9.0f}\n")
        out.write("\n")
    t = per_kernel(f"{src}/grbm.csv")
    out.write("== GRBM_GUI_ACTIVE (GPU-busy cycles per dispatch) ==\n")
    for k, c in sorted(t.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"][0]):
        if "at::native" in k or "rocclr" in k:
            continue
        v, n = c["GRBM_GUI_ACTIVE"]
        out.write(f"{short(k)[:60]:60s} x{n:<4d} {v / n:12.0f}\n")
