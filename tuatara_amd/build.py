"""In-tree build of the native engine for gfx950.

  tuatara_amd/lib/libtuatara_hip.so   HIP kernels + C++ host engine + C ABI (include/tuatara_hip.h)
  build/bindings/pytuatara*.so        pybind11 module (same place the reference puts it,
                                      /root/reference/bindings/run_ocr.py:6)

hipcc cross-compiles without a GPU.  Objects are cached under build/obj and only rebuilt
when a source or header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tuatara_amd", "csrc")
LIBDIR = os.path.join(ROOT, "tuatara_amd", "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
BINDDIR = os.path.join(ROOT, "build", "bindings")
LIB = os.path.join(LIBDIR, "libtuatara_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
SOURCES = ["igemm.hip", "gemm2.hip", "gemm_sp.hip", "split_ops.hip", "attn_split.hip", "attn_cross_split.hip", "conv3p.hip", "conv3h.hip", "conv1u.hip", "qkv_attn4.hip", "conv3s.hip", "gemm_sk.hip", "gemm_skx.hip", "gemm_ws.hip", "mlp_fused.hip", "attn_enc2.hip", "attn_dec2.hip", "qkv_attn.hip", "craft_ops.hip", "parseq_ops.hip", "dec_fused.hip", "post_ops.hip",
           "engine.cpp", "engine_craft.cpp", "engine_parseq.cpp", "engine_pages.cpp", "comm.cpp", "capi.cpp", "capi_debug.cpp", "geometry.cpp", "tuatara.cpp"]
HIP_HOST_SOURCES = {"engine.cpp", "engine_craft.cpp", "engine_parseq.cpp", "engine_pages.cpp", "comm.cpp", "capi.cpp", "capi_debug.cpp"}   # host code that sees HIP types: -x hip
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
COMMON += os.environ.get("TUATARA_EXTRA_HIPCC_FLAGS", "").split()   # experiment builds (e.g. -DMLP_ABLATE_BUILDS); touch the source to rebuild


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return hs


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _headers()):
        cmd = [HIPCC] + COMMON + (["-x", "hip"] if src in HIP_HOST_SOURCES else []) + ["-c", path, "-o", obj]
        subprocess.check_call(cmd)
    return obj


def build_lib(verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl", "-L/opt/rocm/lib", "-lrccl"])
    return LIB


def build_pytuatara() -> str:
    import pybind11

    os.makedirs(BINDDIR, exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(BINDDIR, "pytuatara" + ext)
    src = os.path.join(ROOT, "bindings", "python.cpp")
    if _stale(out, [src, LIB] + _headers()):
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
               src, "-o", out, "-L", LIBDIR, "-ltuatara_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,$ORIGIN/../../tuatara_amd/lib"]
        subprocess.check_call(cmd)
    return out


def build_examples() -> str:
    """build/examples/ocr_cli: counterpart of the reference's examples/resume.cpp / table.cpp (own PNG reader, zlib)."""
    outdir = os.path.join(ROOT, "build", "examples")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "ocr_cli")
    srcs = [os.path.join(ROOT, "examples", "ocr_cli.cpp"), os.path.join(ROOT, "examples", "png_decode.h")]
    if _stale(out, srcs + [LIB] + _headers()):
        cmd = ["g++", "-O2", "-std=c++17", srcs[0], "-o", out, "-L", LIBDIR, "-ltuatara_hip", "-lz", f"-Wl,-rpath,{LIBDIR}",
               "-Wl,-rpath,$ORIGIN/../../tuatara_amd/lib"]
        subprocess.check_call(cmd)
    return out


def stamp_build() -> None:
    """.build_hash = the commit the library was built from (profile scripts on the GPU box, which has no .git, quote it)."""
    try:
        h = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(["git", "diff", "--quiet", "HEAD", "--", "tuatara_amd/csrc"], cwd=ROOT) != 0
        with open(os.path.join(ROOT, ".build_hash"), "w") as f:
            f.write(h + ("+" if dirty else "") + "\n")
    except Exception:
        pass


def build_all() -> None:
    build_lib()
    stamp_build()
    build_pytuatara()
    build_examples()


if __name__ == "__main__":
    build_all()
    print(LIB)
