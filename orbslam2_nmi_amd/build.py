"""Builds orbslam2_nmi_amd/lib/libnmi_hip.so (gfx950) in-tree with hipcc.

The shared library is the product: HIP kernels (csrc/nmi_kernels.hip scoring, csrc/nmi_producers.hip warp / render
stacks) + the C ABI of include/nmi_hip.h (csrc/nmi_capi*.cpp over csrc/nmi_ctx.h) + the host-side mirror of the
reference's driver types (host/*.cpp, include/nmi_host.h).
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, "lib")
# NMI_HIP_LIBRARY points the loader at another build of the same ABI (A/B timing of kernel variants on one box).
LIB = os.environ.get("NMI_HIP_LIBRARY") or os.path.join(LIB_DIR, "libnmi_hip.so")
ARCH = "gfx950"


def sources():
    out = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, sub)
        if os.path.isdir(d):
            out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".cpp"))]
    return out


def headers():
    out = [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, sub)
        if os.path.isdir(d):
            out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".h", ".hpp"))]
    return out


def is_stale():
    if os.environ.get("NMI_HIP_LIBRARY"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in sources() + headers() + [os.path.abspath(__file__)])


# The experiments of profiles/NOTES.md (histogram variants 0 / 2 / 4) are not part of the shipped library; a second
# library with them is built on request:  python -m orbslam2_nmi_amd.build --ablations  -> lib/libnmi_hip_ablate.so, which
# tools/ablate.py loads through NMI_HIP_LIBRARY.
ABLATE_LIB = os.path.join(LIB_DIR, "libnmi_hip_ablate.so")


def flags(ablations=False):
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
            *(["-DNMI_BUILD_ABLATIONS"] if ablations else []),
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-I" + os.path.join(PKG, "host")]


def compile_and_link(out, ablations=False, force=False, verbose=False, jobs=None):
    """One object per source under lib/obj/ (recompiled when the source, any header or -- for kernels -- any .hip file is
    newer), compiled in parallel, then one link.  Same flags for every translation unit; no relocatable device code (no
    kernel calls across units).  Objects and the library are written under private names and renamed, and one builder
    runs at a time (flock): concurrent callers wait and then find nothing left to do."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj_ablate" if ablations else "obj")
    os.makedirs(obj_dir, exist_ok=True)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    def compile_one(job):
        src, obj = job
        tmp = f"{obj}.{os.getpid()}.part"
        run(["hipcc", *flags(ablations), "-c", src, "-o", tmp])
        os.replace(tmp, obj)

    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        newest_header = max(os.path.getmtime(f) for f in headers() + [os.path.abspath(__file__)])
        # some kernel files are compiled more than once through a wrapper that #includes them (nmi_kernels_gated.hip,
        # nmi_kernels_stamped.hip, nmi_pix_kernel.hip include nmi_kernels.hip): every .hip depends on every .hip
        newest_hip = max([os.path.getmtime(f) for f in sources() if f.endswith(".hip")] + [newest_header])
        todo, objs = [], []
        for src in sources():
            obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
            objs.append(obj)
            dep = newest_hip if src.endswith(".hip") else max(os.path.getmtime(src), newest_header)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep:
                todo.append((src, obj))
        if todo:
            with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as ex:
                list(ex.map(compile_one, todo))
        if todo or force or not os.path.exists(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
            tmp = f"{out}.{os.getpid()}.part"
            run(["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp, "-ldl"])
            os.replace(tmp, out)
    return out


def build(force=False, verbose=False):
    """Compile the library if sources are newer than the binary.  Returns the library path."""
    if force or is_stale():
        os.makedirs(LIB_DIR, exist_ok=True)
        compile_and_link(LIB, force=force, verbose=verbose)
    return LIB


if __name__ == "__main__":
    if "--ablations" in sys.argv:
        os.makedirs(LIB_DIR, exist_ok=True)
        compile_and_link(ABLATE_LIB, ablations=True, verbose=True)
        print(ABLATE_LIB)
    else:
        print(build(force="--force" in sys.argv, verbose=True))
