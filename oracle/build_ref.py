"""TEST INFRASTRUCTURE (see oracle/__init__.py): builds oracle/_ref/ -- the REFERENCE's own tracking kernels, for the CPU.

    python oracle/build_ref.py            # needs /root/reference (absent on the GPU box; the built module travels)

/root/reference/src/lib/{droid_kernels,correlation_kernels,altcorr_kernel}.cu + droid.cpp are the CUDA extension
`droid_backends` of the reference (setup.py:9-32).  There is no nvcc and no GPU in the build container, but the kernel
bodies are plain C++ once a dozen CUDA names have a CPU meaning (oracle/ref_build/cuda_on_cpu.h: blocks run one after
the other, the threads of a block are real threads, __syncthreads() is a real barrier).  This script READS the four
files where they lie, rewrites the two things a header cannot express --

  * `kernel<<<grid, block>>>(args)`  ->  `gs_cpu::launch(grid, block, [&]() { kernel(args); })`
  * the six warp-synchronous statements of warpReduce (droid_kernels.cu:35-42) -> read / GS_WARP0_SYNC / write /
    GS_WARP0_SYNC, i.e. what 32 lanes in lockstep do
  * five hand-over points of the one-warp blocks of altcorr_kernel.cu (rewrite_single_warp_lockstep below)

-- writes the rewritten translation units into a temporary directory that is deleted after the compile (no reference
source text enters the repository, tracked or not) and compiles them with g++ against PyTorch's CPU headers into the Python extension module
`oracle/_ref/droid_backends_ref*.so`, whose functions are the reference's own pybind exports (droid.cpp:237-250).
Eigen (an empty submodule in /root/reference) is replaced by oracle/ref_build/include/Eigen/Sparse.

Used by tests/test_reference_kernels_cpu.py to check oracle/droid_oracle.py -- the restatement the GPU parity tests are
judged against -- against the code it restates.  Nothing in go_slam_amd/ or bench.py's timed regions touches it."""
import importlib.util
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = "/root/reference/src/lib"
OUT = os.path.join(HERE, "_ref")
NAME = "droid_backends_ref"
SOURCES = ("droid.cpp", "droid_kernels.cu", "correlation_kernels.cu", "altcorr_kernel.cu")


def _match_back(src, i, open_c, close_c):
    """index of the `open_c` that matches the `close_c` at src[i], scanning backwards"""
    depth = 0
    while i >= 0:
        if src[i] == close_c:
            depth += 1
        elif src[i] == open_c:
            depth -= 1
            if depth == 0:
                return i
        i -= 1
    raise ValueError("unbalanced brackets before a kernel launch")


def _match_fwd(src, i, open_c, close_c):
    depth = 0
    while i < len(src):
        if src[i] == open_c:
            depth += 1
        elif src[i] == close_c:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced brackets after a kernel launch")


def rewrite_launches(src):
    out, pos, n = [], 0, 0
    while True:
        p = src.find("<<<", pos)
        if p < 0:
            out.append(src[pos:])
            return "".join(out), n
        q = src.index(">>>", p)
        cfg = src[p + 3:q].strip()
        # the kernel expression in front of `<<<`: identifier, optionally with template arguments
        k = p - 1
        while src[k].isspace():
            k -= 1
        if src[k] == ">":
            k = _match_back(src, k, "<", ">") - 1
        while k >= 0 and (src[k].isalnum() or src[k] in "_:"):
            k -= 1
        kernel = src[k + 1:p].strip()
        a0 = q + 3
        while src[a0].isspace():
            a0 += 1
        assert src[a0] == "(", "a kernel launch must be followed by its argument list"
        a1 = _match_fwd(src, a0, "(", ")")
        args = src[a0 + 1:a1]
        out.append(src[pos:k + 1])
        # the arguments are evaluated ONCE, in the launching thread (as a CUDA launch does: an accessor of the wrong rank
        # must raise here, not inside a worker thread), and every thread of the grid receives copies of them
        out.append(f"{{ auto gs_args = std::make_tuple({args}); "
                   f"gs_cpu::launch({cfg}, [&]() {{ std::apply({kernel}, gs_args); }}); }}")
        pos = a1 + 1
        n += 1


def rewrite_warp_reduce(src):
    m = re.search(r"void\s+warpReduce\s*\(", src)
    if not m:
        return src, 0
    b0 = src.index("{", m.end())
    b1 = _match_fwd(src, b0, "{", "}")
    body = src[b0:b1 + 1]
    body, n = re.subn(r"sdata\[tid\]\s*\+=\s*sdata\[tid\s*\+\s*(\d+)\];",
                      r"{ const float gs_v = sdata[tid + \1]; GS_WARP0_SYNC(); sdata[tid] += gs_v; GS_WARP0_SYNC(); }", body)
    return src[:b0] + body + src[b1 + 1:], n


def rewrite_single_warp_lockstep(src):
    """altcorr_kernel.cu launches blocks of 4 x 8 = 32 threads -- ONE warp -- and relies on its lanes running in lockstep:
    a lane writes x2s[tid] / y2s[tid] and every lane then reads the OTHER lanes' entries without a barrier
    (altcorr_kernel.cu:71-88, :211-226), and the next (iy, ix) step overwrites the f2 tile the other lanes have just read
    (:89-104).  A cooperative schedule needs those hand-over points spelled out; each insertion below is a
    __syncthreads() of the one-warp block at a point every lane reaches (the kernels have no early return)."""
    sync = " GS_LOCKSTEP();"
    n = 0
    a = src.index("void altcorr_forward_kernel")
    b = src.index("void altcorr_backward_kernel")
    c = src.index("altcorr_cuda_forward")
    fwd, bwd = src[a:b], src[b:c]
    fwd, k = re.subn(r"(y2s\[tid\] = coords\[b\]\[n\]\[h1\]\[w1\]\[1\];\s*\})", r"\1" + sync, fwd)
    n += k
    fwd, k = re.subn(r"(s \+= f1\[k\]\[tid\] \* f2\[k\]\[tid\];)", r"\1" + sync, fwd)
    n += k
    bwd, k = re.subn(r"(y2s\[tid\] = coords\[b\]\[n\]\[h1\]\[w1\]\[1\];)", r"\1" + sync, bwd)
    n += k
    bwd, k = re.subn(r"(f2_grad\[k\]\[tid\] \+= g \* f1\[k\]\[tid\];\s*\})", r"\1" + sync, bwd)
    n += k
    bwd, k = re.subn(r"(atomicAdd\(fptr\+c\+c2, f2_grad\[c2\]\[k1\]\);\s*\})", r"\1" + sync, bwd)
    n += k
    return src[:a] + fwd + bwd + src[c:], n


def module_path():
    if not os.path.isdir(OUT):
        return None
    for f in sorted(os.listdir(OUT)):
        if f.startswith(NAME) and f.endswith(".so"):
            return os.path.join(OUT, f)
    return None


def load():
    """the built module, or None if it does not exist (no /root/reference and nothing prebuilt)"""
    path = module_path()
    if path is None:
        return None
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force=False, verbose=False):
    """compile oracle/_ref/droid_backends_ref*.so from the reference's sources; returns its path, or None when
    /root/reference is absent (the GPU box: a prebuilt module, if any, is used as it is)"""
    if not os.path.isdir(REF_LIB):
        return module_path()
    if module_path() and not force:
        newest = max(os.path.getmtime(os.path.join(REF_LIB, s)) for s in SOURCES)
        mine = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(HERE, "ref_build"))
                   for f in fs)
        mine = max(mine, os.path.getmtime(os.path.abspath(__file__)))
        if os.path.getmtime(module_path()) > max(newest, mine):
            return module_path()
    import shutil
    import tempfile
    from torch.utils import cpp_extension
    # the rewritten translation units are build intermediates: they live in a temporary directory for the duration of the
    # compile and are deleted with it -- only the binary module is kept (oracle/_ref/, git-ignored)
    bdir = tempfile.mkdtemp(prefix="goslam_ref_build_")
    try:
        sdir = os.path.join(bdir, "src")
        os.makedirs(sdir)
        srcs, stats = [], {}
        for s in SOURCES:
            text = open(os.path.join(REF_LIB, s)).read()
            text, n_l = rewrite_launches(text)
            text, n_w = rewrite_warp_reduce(text)
            n_s = 0
            if s == "altcorr_kernel.cu":
                text, n_s = rewrite_single_warp_lockstep(text)
            stats[s] = (n_l, n_w, n_s)
            dst = os.path.join(sdir, os.path.splitext(s)[0] + "_cpu.cpp")
            open(dst, "w").write(text)
            srcs.append(dst)
        assert stats["droid_kernels.cu"][1] == 6, f"warpReduce changed shape: {stats}"
        assert stats["altcorr_kernel.cu"][2] == 5, f"altcorr kernels changed shape: {stats}"
        shim = os.path.join(HERE, "ref_build")
        cpp_extension.load(name=NAME, sources=srcs, build_directory=bdir, with_cuda=False, verbose=verbose,
                           extra_include_paths=[os.path.join(shim, "include")], is_python_module=False,
                           extra_cflags=["-O2", "-std=c++17", "-include", os.path.join(shim, "cuda_on_cpu.h"), "-w"])
        os.makedirs(OUT, exist_ok=True)
        for f in os.listdir(bdir):
            if f.startswith(NAME) and f.endswith(".so"):
                shutil.copy2(os.path.join(bdir, f), os.path.join(OUT, f))
    finally:
        shutil.rmtree(bdir, ignore_errors=True)
    return module_path()


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
