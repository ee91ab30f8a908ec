#!/usr/bin/env python3
"""In-tree ahead-of-time build of the native extensions (no JIT cache, no hipify pass).

Two shared objects are produced inside the package directory so that they travel with
the source tree:

* ``distributed_pytorch_amd/_host*.so`` -- g++ / pybind11 only: TCP rendezvous store and
  bucket planner.  Needs neither torch headers nor the HIP toolchain.
* ``distributed_pytorch_amd/_C*.so`` -- hipcc ``--offload-arch=gfx950``: the HIP kernels
  (GEMM, softmax cross-entropy, AdamW, bucket pack), the RCCL communicator and the
  gradient reducer, linked against the HIP / RCCL runtime that ``import torch`` loads.

Usage: ``python build.py [--host-only] [--force] [-j N] [--verbose]``.
Objects are cached under ``build/`` and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shlex
import subprocess
import sys
import sysconfig
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
PKG = ROOT / "distributed_pytorch_amd"
BUILD = ROOT / "build"
GPU_ARCH = "gfx950"
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def _run(cmd, verbose):
    if verbose:
        print("  $", " ".join(shlex.quote(str(c)) for c in cmd), flush=True)
    proc = subprocess.run([str(c) for c in cmd], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError(f"command failed ({proc.returncode}): {' '.join(map(str, cmd))}")
    if verbose and proc.stdout.strip():
        print(proc.stdout)
    return proc.stdout


def _newest_header_mtime():
    newest = 0.0
    for p in CSRC.rglob("*"):
        if p.suffix in (".h", ".hpp", ".cuh", ".inc"):
            newest = max(newest, p.stat().st_mtime)
    return newest


def _needs_build(src: Path, obj: Path, header_mtime: float, force: bool) -> bool:
    if force or not obj.exists():
        return True
    t = obj.stat().st_mtime
    return src.stat().st_mtime > t or header_mtime > t or Path(__file__).stat().st_mtime > t


def _python_includes():
    import pybind11

    return ["-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include()]


def build_host(force=False, verbose=False):
    """g++ build of the torch-free, GPU-free module."""
    out = PKG / f"_host{EXT_SUFFIX}"
    srcs = [CSRC / "host_bindings.cpp", CSRC / "store" / "tcp_store.cpp", CSRC / "reducer" / "bucket_plan.cpp"]
    hdr = _newest_header_mtime()
    objs = []
    jobs = []
    (BUILD / "host").mkdir(parents=True, exist_ok=True)
    for s in srcs:
        o = BUILD / "host" / (s.stem + ".o")
        objs.append(o)
        if _needs_build(s, o, hdr, force):
            jobs.append(["g++", "-O2", "-g0", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-I" + str(CSRC), *_python_includes(), "-c", s, "-o", o])
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    if jobs or not out.exists() or any(o.stat().st_mtime > out.stat().st_mtime for o in objs):
        _run(["g++", "-shared", "-o", out, *objs, "-lpthread"], verbose)
    return out


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    tdir = Path(torch.__file__).resolve().parent
    incs = [str(p) for p in ce.include_paths(device_type="cuda")] if "device_type" in ce.include_paths.__code__.co_varnames else [str(p) for p in ce.include_paths()]
    return tdir, incs


def _gpu_sources():
    kernels = sorted((CSRC / "kernels").glob("*.hip"))
    torch_cpp = [p for p in [CSRC / "bindings.cpp", CSRC / "comm" / "rccl_comm.cpp", CSRC / "comm" / "comm_stream.cpp", CSRC / "reducer" / "reducer.cpp", CSRC / "ops_bindings.cpp"] if p.exists()]
    plain_cpp = [CSRC / "store" / "tcp_store.cpp", CSRC / "reducer" / "bucket_plan.cpp"]
    return kernels, torch_cpp, plain_cpp


def build_gpu(force=False, verbose=False, jobs_n=8):
    """hipcc build of the HIP kernels + RCCL communicator + reducer for gfx950."""
    rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
    hipcc = rocm / "bin" / "hipcc"
    tdir, tincs = _torch_paths()
    out = PKG / f"_C{EXT_SUFFIX}"
    kernels, torch_cpp, plain_cpp = _gpu_sources()
    hdr = _newest_header_mtime()
    (BUILD / "gpu").mkdir(parents=True, exist_ok=True)

    common = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={GPU_ARCH}", "-I" + str(CSRC), "-I" + str(rocm / "include")]
    kernel_flags = common + ["-fno-math-errno", "-ffp-contract=fast", "-fno-gpu-rdc", "-munsafe-fp-atomics", "-Wno-unused-result"]
    torch_flags = common + [
        "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-DTORCH_EXTENSION_NAME=_C",
        "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=1", "-DUSE_C10D_GLOO", "-DUSE_DISTRIBUTED",
        "-Wno-deprecated-declarations", "-Wno-unused-result", "-fno-gpu-rdc",
        # torch's bundled pybind11 must win over any other copy: its ABI tag is what makes
        # torch's registered types (tensors, c10d process groups) castable in this module.
        *["-isystem" + i for i in tincs], "-I" + sysconfig.get_paths()["include"],
    ]
    plain_flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + str(CSRC)]

    # Per-file additions.  attention.hip: its maxima run over raw matrix-core outputs (no NaN can occur:
    # masked scores are -inf, which these flags leave alone), and without -fno-honor-nans every fmaxf gets a
    # canonicalising v_max in front; SLP packing of adjacent f32 operations into v_pk_*_f32 is off because the
    # packed forms issue slower than two plain instructions next to MFMAs.
    # -amdgpu-mfma-vgpr-form: the one-wave-per-SIMD forward kernel has a 512-register budget, with which the
    # compiler would give every matrix instruction an AGPR destination and copy the score accumulators to
    # ordinary registers for the softmax; the file's other kernels (256-register budget) already get this form.
    per_file = {"attention.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]}
    objs, cmds = [], []
    draw_asm = None
    for s in kernels:
        o = BUILD / "gpu" / (s.stem + ".hip.o")
        objs.append(o)
        if _needs_build(s, o, hdr, force):
            cmds.append([hipcc, *kernel_flags, *per_file.get(s.name, []), "-c", s, "-o", o])
            if s.name == "gemm_bf16.hip":
                # the same translation unit once more as assembly, for the register check below
                draw_asm = BUILD / "gpu" / "gemm_bf16.s"
                cmds.append([hipcc, *kernel_flags, "--cuda-device-only", "-S", s, "-o", draw_asm])
    for s in torch_cpp:
        o = BUILD / "gpu" / (s.stem + ".cpp.o")
        objs.append(o)
        if _needs_build(s, o, hdr, force):
            cmds.append([hipcc, "-x", "hip", *torch_flags, "-c", s, "-o", o])
    for s in plain_cpp:
        o = BUILD / "gpu" / (s.stem + ".plain.o")
        objs.append(o)
        if _needs_build(s, o, hdr, force):
            cmds.append(["g++", *plain_flags, "-c", s, "-o", o])

    t0 = time.time()
    with cf.ThreadPoolExecutor(max_workers=max(1, jobs_n)) as ex:
        list(ex.map(lambda c: _run(c, verbose), cmds))
    if draw_asm is not None:
        # The streaming GEMM consumes the result of a returning atomic behind a counted wait, both in inline
        # assembly: a build in which the compiler copies or spills the destination register in between reads a
        # stale tile id (tools/check_draw_isa.py has the details).  Such a build must not ship.
        sys.path.insert(0, str(ROOT / "tools"))
        try:
            import check_draw_isa
        finally:
            sys.path.pop(0)
        problems = check_draw_isa.check(draw_asm.read_text())
        if problems:
            (BUILD / "gpu" / "gemm_bf16.hip.o").unlink(missing_ok=True)
            raise RuntimeError("gemm_bf16.hip: the compiler touches the drawn tile id before it has arrived:\n  " + "\n  ".join(problems))
        print("[build] gemm_bf16: drawn-tile-id register check ok", flush=True)
    stale = (not out.exists()) or any(o.stat().st_mtime > out.stat().st_mtime for o in objs)
    if cmds or stale:
        tlib = tdir / "lib"
        _run([
            hipcc, "-shared", f"--offload-arch={GPU_ARCH}", "-fno-gpu-rdc", "-o", out, *objs,
            "-L" + str(tlib), "-Wl,-rpath," + str(tlib), "-Wl,--no-as-needed",
            "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-l:librccl.so", "-l:libamdhip64.so", "-lpthread", "-ldl",
        ], verbose)
    if verbose or cmds:
        print(f"[build] _C: {len(cmds)} object(s) rebuilt in {time.time() - t0:.1f}s -> {out.name}", flush=True)
    return out


def build_all(force=False, verbose=False, host_only=False, jobs=8):
    outs = [build_host(force=force, verbose=verbose)]
    if not host_only:
        outs.append(build_gpu(force=force, verbose=verbose, jobs_n=jobs))
    return outs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--host-only", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("-j", type=int, default=min(8, os.cpu_count() or 4))
    a = ap.parse_args()
    for o in build_all(force=a.force, verbose=a.verbose, host_only=a.host_only, jobs=a.j):
        print("built", o)


if __name__ == "__main__":
    main()
