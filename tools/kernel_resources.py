#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in csrc/<name>.o (from the code object's metadata notes).
    python tools/kernel_resources.py mlp_coop mlp_layer"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(name):
    obj = os.path.join(ROOT, "3dsad-main_amd", "csrc", name + ".o")
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "x.fat"), os.path.join(td, "x.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                        vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                        lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size")))
    return out


if __name__ == "__main__":
    for n in sys.argv[1:]:
        for k in kernels(n):
            dem = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            print(f"{dem[:70]:70s} vgpr {k['vgpr']:>4s} agpr {k['agpr']:>3s} sgpr {k['sgpr']:>4s} vspill {k['vspill']:>3s} "
                  f"sspill {k['sspill']:>3s} lds {k['lds']:>6s} scratch {k['scratch']:>5s}")
