#!/usr/bin/env python3
"""Copy the judged summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked).

usage: tools/collect_profiles.py <tag> [round]     e.g. tools/collect_profiles.py r01c r01
Writes profiles/<round>_bench_n1.json.log, <round>_bench_kernel_stats.csv, <round>_pmc_summary.txt and
profiles/traffic_latest.json (k_align HBM-side bytes per launch, corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE is KiB and under-reports by 2x on gfx950 -> 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024).
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench_n1.json.log"), os.path.join(dst, rnd + "_bench_n1.json.log"))
shutil.copy(os.path.join(src, "kernel_stats_nimble.csv"), os.path.join(dst, rnd + "_bench_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench_n1.json.log")).read().strip().splitlines()[-1])
summary = open(os.path.join(src, "pmc", "summary.txt")).read()
head = ("# rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0\n"
        "# (tools/profile_round.sh; one pass per counter group, never combined with sys/hip traces)\n"
        "# MI355X, workload: %s\n"
        "# averages per dispatch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles;\n"
        "# FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE tallies 128-B requests at 64 B on gfx950 (double it).\n"
        % bench["config"]["workload"])
open(os.path.join(dst, rnd + "_pmc_summary.txt"), "w").write(head + summary)
# the align stage of a call: one launch (k_align<false, false, false[, 0]>) or, since round 4, the fast launch and the launch that
# redoes what it left (k_align<..., 1> and <..., 2>): their counters add up to the stage's
vals, launches = {}, []
for name in ("k_align<false, false, false>", "k_align<false, false, false, 0>", "k_align<false, false, false, 1>",
             "k_align<false, false, false, 2>"):
    blk = re.search(r"^" + re.escape(name) + r"\n((?:  .*\n)+)", summary, re.M)
    if not blk:
        continue
    launches.append(name)
    for k, v in re.findall(r"^\s+(\S+)\s+avg/dispatch\s+(\S+)", blk.group(1), re.M):
        vals[k] = vals.get(k, 0.0) + float(v)
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    fetch, write = float(vals["FETCH_SIZE"]) * 1024, float(vals["WRITE_SIZE"]) * 1024
    t = {"reads": bench["config"]["reads_per_gpu"], "features": bench["config"]["features"],
         "workload": bench["config"].get("workload_id", "configs2"),
         "k_align_hbm_bytes_per_launch": 2 * fetch + write,
         "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
         "correction": "2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE; memory-side of L2, "
                       "Infinity-Cache hits included",
         "launches_summed": launches,
         "sq_insts_valu": vals.get("SQ_INSTS_VALU"), "tcc_miss": vals.get("TCC_MISS_sum"), "tcc_hit": vals.get("TCC_HIT_sum"),
         "source": "profiles/%s_pmc_summary.txt" % rnd}
    json.dump(t, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(t))
