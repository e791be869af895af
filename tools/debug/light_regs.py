#!/usr/bin/env python3
"""tools/debug/light_regs.py [H] [MINW] [--all] -- registers / LDS / occupancy of the tick launch compiled for ONE body type at a time (or for
the light kernel's whole type mask): what each light body needs on its own, before a shared __launch_bounds__ squeezes it.
Compiles a probe translation unit against beatrice-vst_amd/csrc with -Rpass-analysis=kernel-resource-usage."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TYPES = ["T_F1", "T_FFT", "T_HEAD", "T_COND", "T_VQ", "T_F2L", "T_F3L", "T_P23L", "T_POUTL", "T_OUTL", "T_INPL", "T_UP1L", "T_RES1AL", "T_RES1BL", "T_UP2L"]
SRC = r'''
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <deque>
#include <numeric>
#include <random>
#include <string>
#include <type_traits>
#include <chrono>
#include <vector>
#include "abi_objects.h"
#include "beatrice_batch.h"
#include "tick.hip.h"
namespace fuse {
template <int MINW, unsigned long long MASK, class... Ms> const void* kptr(const Table<Ms...>*) { return (const void*)table_kernel_w<MINW, false, MASK, Ms...>; }
}
#define ONE(T) fuse::kptr<PROBE_MINW, tick::bit(tick::T)>((const tick::Ops<PROBE_H>::Tab*)nullptr)
const void* probes[] = { PROBE_LIST };
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    H = int(args[0]) if args else 4
    minw = int(args[1]) if len(args) > 1 else 1
    whole = "--all" in sys.argv
    lst = "fuse::kptr<PROBE_MINW, tick::kLightTypes>((const tick::Ops<PROBE_H>::Tab*)nullptr)" if whole else ", ".join("ONE(%s)" % t for t in TYPES)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.hip")
        open(src, "w").write(SRC)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I" + os.path.join(REPO, "include"),
               "-I" + os.path.join(REPO, "beatrice-vst_amd", "csrc"), "-DPROBE_H=%d" % H, "-DPROBE_MINW=%d" % minw, "-DPROBE_LIST=" + lst,
               "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "probe.o")] + [a for a in sys.argv[1:] if a.startswith("--D") and False]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
    # enum values of the types, from tick.hip.h
    enum = re.search(r"enum BodyType \{(.*?)\};", open(os.path.join(REPO, "beatrice-vst_amd", "csrc", "tick.hip.h")).read(), re.S).group(1)
    names = [n.strip() for n in enum.replace("\n", " ").split(",") if n.strip()]
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print("H = %d, __launch_bounds__(512, %d)" % (H, minw))
    for name, v in rows.items():
        m = re.match(r"_ZN4fuse14table_kernel_wILi\d+ELb0ELy(\d+)E", name)
        if not m:
            continue
        mask = int(m.group(1))
        label = "light kernel (all light types)" if bin(mask).count("1") > 1 else names[mask.bit_length() - 1]
        print("%-32s VGPRs %3d AGPRs %3d  spill %3d  scratch %5d B  LDS %6d B  occupancy %d waves/SIMD" % (
            label, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1), v.get("ScratchSize [bytes/lane]", -1), v.get("LDS Size [bytes/block]", -1), v.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
