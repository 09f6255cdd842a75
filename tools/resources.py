"""hipcc -Rpass-analysis=kernel-resource-usage remarks (stderr text) -> one line per kernel: VGPRs, scratch, occupancy, LDS.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -c query_amd/csrc/n1k_kernels.hip -o /tmp/k.o \
          -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt && python tools/resources.py /tmp/res.txt [substring ...]
"""
import re, subprocess, sys

t = open(sys.argv[1]).read()
want = sys.argv[2:]
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    if want and not any(w in n for w in want):
        continue
    print("%-120s vgpr=%d agpr=%d sgpr=%d scratch=%d occ=%d lds=%d" % (n[:120], g("VGPRs"), g("AGPRs"), g("SGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
