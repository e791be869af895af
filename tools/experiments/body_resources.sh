#!/bin/bash
# Registers / spills of every tick body compiled as a kernel of its own (the table kernel reports one figure for all of them):
#   tools/experiments/body_resources.sh
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -S -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -I include -I beatrice-vst_amd/csrc \
  tools/experiments/body_resources.hip -o /tmp/body_resources.s || exit 1
python3 - <<'PY'
import re, subprocess
txt = open('/tmp/body_resources.s').read()
meta = txt[txt.index('amdhsa.kernels:'):]
rows = []
for blk in meta.split('  - .agpr_count')[1:]:
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
    rows.append((g('vgpr_spill_count'), g('vgpr_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), re.search(r'\.name:\s+(\S+)', blk).group(1)))
dem = subprocess.run(['c++filt'], input='\n'.join(r[4] for r in rows), capture_output=True, text=True).stdout.split('\n')
for (sp, v, sg, pr, n), d in sorted(zip(rows, dem), reverse=True):
    if 'body_kernel' in d:
        print('vgpr spills %3d  vgprs %3d  sgpr spills %3d  scratch %4d  %s' % (sp, v, sg, pr, d.replace('void body_kernel<', '')[:110]))
PY
