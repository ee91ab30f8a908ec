"""Checks the compiled bf16 GEMM for a hazard the compiler cannot see.

The streaming GEMM kernel draws its next tile id with a returning ``global_atomic_add`` issued from inline assembly
and consumes the value a whole k-step later, behind a counted ``s_waitcnt vmcnt(6)`` that is inline assembly as well.
To the compiler the destination register holds the result as soon as the statement has been passed, so it is free
to copy or reuse that register before the data has arrived; the hardware has no interlock for that.  This script
compiles ``csrc/kernels/gemm_bf16.hip`` to gfx950 assembly and asserts that, for every such atomic, no instruction
between it and the first covering wait (``vmcnt(n)``, n <= 6) names the destination register.

    python tools/check_draw_isa.py            # compile + check, exit status 1 on a violation
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ATOMIC = re.compile(r"^\s*global_atomic_add\s+v(\d+)\s*,")
_WAIT = re.compile(r"s_waitcnt\s+vmcnt\((\d+)\)")
_SINGLE = re.compile(r"\bv(\d+)\b")
_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
MAX_LINES = 4000  # a k-step is ~600 lines of assembly


def _names_register(line: str, reg: int) -> bool:
    code = line.split(";", 1)[0]
    if any(int(m.group(1)) == reg for m in _SINGLE.finditer(code)):
        return True
    return any(int(m.group(1)) <= reg <= int(m.group(2)) for m in _RANGE.finditer(code))


def check(asm: str) -> List[str]:
    """Returns one message per violation (empty list: the build is safe); raises if no atomic is found at all."""
    lines = asm.splitlines()
    problems: List[str] = []
    atomics = 0
    for i, line in enumerate(lines):
        m = _ATOMIC.match(line)
        if not m:
            continue
        atomics += 1
        reg = int(m.group(1))
        verdict = f"line {i + 1}: no covering s_waitcnt vmcnt(<=6) within {MAX_LINES} lines of the atomic"
        for j in range(i + 1, min(len(lines), i + MAX_LINES)):
            w = _WAIT.search(lines[j])
            if w and int(w.group(1)) <= 6:
                verdict = ""
                break
            if _names_register(lines[j], reg):
                verdict = f"line {j + 1}: `{lines[j].strip()}` touches v{reg} before the wait that covers the atomic of line {i + 1}"
                break
        if verdict:
            problems.append(verdict)
    if atomics == 0:
        raise RuntimeError("no global_atomic_add found: is this the bf16 GEMM's assembly?")
    return problems


def compile_to_asm(out_path: str) -> None:
    # the kernel flags of build.py: register allocation, and with it the outcome of this check, depends on them
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", f"-I{ROOT}/csrc", "-I/opt/rocm/include",
           "-fno-math-errno", "-ffp-contract=fast", "-fno-gpu-rdc", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
           f"{ROOT}/csrc/kernels/gemm_bf16.hip", "-o", out_path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main() -> int:
    if len(sys.argv) > 1:
        asm = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "gemm_bf16.s")
            compile_to_asm(path)
            asm = open(path).read()
    problems = check(asm)
    for p in problems:
        print(p)
    print("draw-id register check:", "FAILED" if problems else "ok")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
