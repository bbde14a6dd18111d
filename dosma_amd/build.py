"""Build the HIP shared library (libqmri_hip.so) in-tree for gfx950.

    python -m dosma_amd.build            # hipcc --offload-arch=gfx950 ... -> dosma_amd/libqmri_hip.so
    python -m dosma_amd.build --clean    # drop experiment variants / probe binaries (what gpurun would push for nothing)

The .so is git-ignored but travels with the gpurun snapshot.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libqmri_hip.so")
SOURCES = ["qmri_capi.hip", "monoexp_lm.hip", "linfit.hip", "lm_generic.hip", "unet_kernels.hip", "unet_rw.hip", "unet_s3.hip", "unet_c4.hip", "unet_d4.hip", "unet_enc0.hip", "unet_engine.hip", "dess.hip", "region_stats.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "qmri.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, extra_flags) -> str:
    """A second library next to the product one (dosma_amd/libqmri_hip_<name>.so) compiled with extra hipcc flags --
    used for timing experiments (-DQMRI_S3_EXPERIMENTS); selected at run time with DOSMA_AMD_LIB=<path>."""
    bdir = os.path.join(HERE, "build", name)
    os.makedirs(bdir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)] + [os.path.getmtime(os.path.join(ROOT, "include", "qmri.h"))]):
            continue
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *extra_flags,
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    out = os.path.join(HERE, f"libqmri_hip_{name}.so")
    subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    # one builder at a time (several ranks of a torchrun job may find the library missing together): the others wait
    # on the lock and then find a fresh library
    import fcntl

    with open(os.path.join(bdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return SO
        return _build_locked(bdir, verbose, force)


def _build_locked(bdir: str, verbose: bool, force: bool = False) -> str:
    objs = []
    procs = []
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "qmri.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        # per-file staleness: an object newer than its source and every header is reused (a kernel edit recompiles one
        # translation unit, not nine)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(os.path.join(CSRC, src)), hdr_time)):
            continue
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    tmp = SO + f".tmp{os.getpid()}"
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, SO)  # atomic: a concurrent loader never sees a half-written library
    return SO


def clean(verbose: bool = True) -> list:
    """Remove everything a timing experiment leaves behind and `gpurun` would otherwise push to the GPU box: the variant
    object trees (dosma_amd/build/<name>/), the variant libraries (libqmri_hip_<name>.so) and the probe binaries under
    scripts/probes/ (their .hip sources stay).  The product library and its objects are kept."""
    removed = []
    bdir = os.path.join(HERE, "build")
    if os.path.isdir(bdir):
        for name in sorted(os.listdir(bdir)):
            d = os.path.join(bdir, name)
            if os.path.isdir(d):
                shutil.rmtree(d)
                removed.append(d)
    for name in sorted(os.listdir(HERE)):
        if name.startswith("libqmri_hip_") and name.endswith(".so"):
            os.remove(os.path.join(HERE, name))
            removed.append(os.path.join(HERE, name))
    probes = os.path.join(ROOT, "scripts", "probes")
    if os.path.isdir(probes):
        for name in sorted(os.listdir(probes)):
            f = os.path.join(probes, name)
            if os.path.isfile(f) and not name.endswith(".hip"):
                os.remove(f)
                removed.append(f)
    for name in sorted(os.listdir(os.path.join(ROOT, "scripts"))):
        if name.endswith(".so"):
            os.remove(os.path.join(ROOT, "scripts", name))
            removed.append(os.path.join(ROOT, "scripts", name))
    if verbose:
        for r in removed:
            print("removed", os.path.relpath(r, ROOT))
    return removed


if __name__ == "__main__":
    if "--clean" in sys.argv:
        clean()
    elif "--experiments" in sys.argv:
        print(build_variant("exp", ["-DQMRI_S3_EXPERIMENTS"]))
    elif "--variant" in sys.argv:  # python -m dosma_amd.build --variant NAME -DFLAG ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
