"""Wave clock cycles per section of k_align (needs a build with -DNIMBLE_PROFILE_SECTIONS=1: tools/build_variant.sh prof
-DNIMBLE_PROFILE_SECTIONS=1, run with NIMBLE_LIB_DIR=.../libv/prof).  python tools/section_probe.py [T] [N]"""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
names, seqs = synth.make_library(T)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
ctx = lib.device_context()
SECT = ["tile fetch+barrier", "key load+probe", "partition", "walk misc", "seed phase", "walk loop", "class+thresholds",
        "stores+tail", "re-seed search", "-"]
CASES = [
    ("bench recipe", None, 0.005),
    ("on-target, exact", (1.0, 1.0, 1.0, 1.0), 0.0),
    ("on-target, 0.5 % substitutions", (1.0, 1.0, 1.0, 1.0), 0.005),
    ("off-target", (0.0, 1.0, 1.0, 1.0), 0.0),
    ("low complexity (prefiltered)", (0.0, 0.0, 0.0, 1.0), 0.0),
    ("exact, all reads identical", (1.0, 1.0, 1.0, 1.0), 0.0),
]
out = (C.c_uint64 * 16)()
for tag, mix, subst in CASES:
    reads = synth.make_reads_torch(seqs, N, L=150, seed=synth.READ_SEED, device="cuda:0", mix=mix, subst=subst)
    if "identical" in tag:
        reads = reads[:1].expand(N, -1).contiguous()
    torch.cuda.synchronize()
    for rep in range(3):
        if rep == 2:
            nim.hip_lib().nimble_debug_sections(out, 1)
        lib.score_call_raw(reads, None, n=N, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
        t = ctx.timing()
    if nim.hip_lib().nimble_debug_sections(out, 1) != 1:
        raise SystemExit("this build has no section clocks")
    if os.environ.get("SECTION_RAW"):   # a -DNIMBLE_PROFILE_SECTIONS=2 build: wave-level trip counts of the divergent blocks
        names = ["tiles", "walk_fast calls", "walk iterations", "record loads", "mismatch blocks", "commits", "junctions",
                 "find_match calls", "scan rounds", "filter-line blocks", "candidate probes", "re-seed searches",
                 "left-extension steps", "class lookups", "intern probe steps", "dictionary second probes"]
        print("%-32s k_align %.3f ms  " % (tag, t["align"]) + "  ".join("%s %d" % (names[i], out[i]) for i in range(16)), flush=True)
        del reads
        continue
    tot = float(sum(out[i] for i in range(9))) or 1.0
    print("%-32s k_align %.3f ms  " % (tag, t["align"]) + "  ".join("%s %.1f%%" % (SECT[i], 100.0 * out[i] / tot) for i in range(9)), flush=True)
    del reads
