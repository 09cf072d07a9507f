#!/bin/bash
# VGPR / spill / scratch of the kernels of one apply_inst translation unit:  tools/kernel_resources.sh double 4 [name filter]
T=${1:-double}; P=${2:-4}; F=${3:-persistent}
cd /tmp && [ -n "$SKIP_COMPILE" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I/root/repo/include -DMGAMD_INST_T=$T -DMGAMD_INST_P=$P \
  -Rpass-analysis=kernel-resource-usage -c /root/repo/dealii_multigrid_amd/csrc/apply_inst.hip -o /tmp/kres_$T$P.o 2> /tmp/kres_$T$P.txt
python3 - "$T" "$P" "$F" <<'PY'
import re, sys, subprocess
T, P, F = sys.argv[1:4]
txt = open(f'/tmp/kres_{T}{P}.txt').read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if F not in dem:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, '?'])[1]
    scr, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]")
    print("%-100s VGPR %3s spill %3s scratch %4s SGPR %3s LDS %s" % (dem.split('(')[0][:100], g('VGPRs'), g('VGPRs Spill'), scr, g('SGPRs'), lds))
PY
