#!/usr/bin/env python3
"""Disassembly of csrc/<name>.o's gfx950 code object:  python tools/kernel_disasm.py mlp_bf16_reg > /tmp/x.s"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1] if sys.argv[1].endswith(".o") else os.path.join(ROOT, "3dsad-main_amd", "csrc", sys.argv[1] + ".o")
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "x.fat"), os.path.join(td, "x.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    sys.stdout.write(subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True))
