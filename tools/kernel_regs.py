"""Diagnostic: VGPR / spill / scratch per kernel of one HIP source (device-only -S compile, gfx950).

  python tools/kernel_regs.py bb-ocr_amd/csrc/conv_mfma.hip [name-filter]
"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(root, "include"),
                "--cuda-device-only", "-S", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
md = s[s.index("amdhsa.kernels"):]
for b in md.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt not in name:
        continue
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
    short = re.sub(r"^_Z\d+", "", name)[:60]
    print(f"{short:60s} vgpr {g('vgpr_count'):4d} spill {g('vgpr_spill_count'):4d} sgpr {g('sgpr_count'):4d} sspill {g('sgpr_spill_count'):3d} "
          f"scratch {g('private_segment_fixed_size'):5d} lds {g('group_segment_fixed_size')}")
print("asm:", out)
