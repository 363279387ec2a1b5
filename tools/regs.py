#!/usr/bin/env python3
"""Per-function register / scratch usage of the gfx950 code object (tuning aid): python tools/regs.py [-DHRG_PHASE='__device__ __noinline__' ...]"""
import glob, os, re, subprocess, sys, tempfile
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "human-robot-gym_amd", "csrc", "hrgym_hip.hip")
d = tempfile.mkdtemp()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-save-temps", "-Wno-unused-value", "-Xarch_device", "-fapprox-func", "-mllvm", "-disable-machine-licm", *sys.argv[1:], "-o", "t.o", src], cwd=d, stderr=subprocess.DEVNULL)
s = open(glob.glob(d + "/*gfx950*.s")[0]).read()
for m in re.finditer(r"\.type\s+(\S+),@function", s):
    name = m.group(1)
    tail = s[m.end():]
    g = lambda k: re.search(r"; %s: (\d+)" % k, tail)
    nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
    code = re.search(r"codeLenInByte = (\d+)", tail).group(1)
    print("%-28s vgpr %4s agpr %4s scratch %5s code %6s" % (nm, g("NumVgprs").group(1), g("NumAgprs").group(1), g("ScratchSize").group(1), code))
m = re.search(r"\.group_segment_fixed_size:\s*(\d+)", s)
print("LDS bytes/workgroup:", m.group(1) if m else "?")
