#!/bin/bash
# usage: tools/kres.sh [pattern]   -- registers / scratch / occupancy of the gfx950 kernels (compile-only, no GPU)
cd "$(dirname "$0")/../nimble-aligner_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $KRES_FLAGS \
  -Rpass-analysis=kernel-resource-usage -c csrc/kernels.hip -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        name = re.sub(r"\(.*", "", name).replace("void nimble::(anonymous namespace)::", "")
        cur = name; rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for n, r in rows.items():
    if pat in n:
        print("%-44s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-5s occ %-2s LDS %s" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
' "$1"
