"""Builds libt2s_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libt2s_hip.so")
SOURCES = ["conv_gemm.hip", "gate_gemm_pp.hip", "wgrad_cl.hip", "waveglow_ops.hip", "tacotron_ops.hip", "sbgemm.hip", "train_ops.hip", "taco_bwd_ops.hip", "t2s_api_taco_bwd.hip", "t2s_api.hip", "t2s_api_taco.hip",
           "t2s_api_train.hip", "audio_ops.hip", "t2s_api_audio.hip", "loss_ops.hip"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "t2s_hip.h")]
    return _newest(deps) > os.path.getmtime(LIB)


def _compile(hipcc, src, obj, verbose):
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
    cmd[1:1] = os.environ.get("T2S_BUILD_DEFINES", "").split()      # e.g. -DT2S_GEMM_ABLATE for the timing-only ablations
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
    return r.stderr


def build(force=False, verbose=False):
    """Compile what is out of date (a source newer than its object, or any header newer than it), a few files at a time, and
    link.  T2S_BUILD_DEFINES changes what an object means, so diagnostic builds always pass force=True (tools/ do)."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(os.path.dirname(HERE), "include", "t2s_hip.h")]
    hdr_t = _newest(headers)
    force = force or bool(os.environ.get("T2S_BUILD_DEFINES", "").strip())
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            jobs.append((src, obj))
    workers = max(1, min(len(jobs), int(os.environ.get("T2S_BUILD_JOBS", "4"))))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            for err in ex.map(lambda so: _compile(hipcc, so[0], so[1], verbose), jobs):
                if verbose:
                    sys.stderr.write(err)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


def build_variant(name, defines):
    """A diagnostic build next to the shipped one: every source compiled with `defines` (e.g. "-DT2S_ATTSTREAM_ABLATE") into
    build/<name>/ and linked to build/<name>/libt2s_hip.so (git-ignored, travels with gpurun).  Load it with T2S_LIB_PATH."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(os.path.dirname(HERE), "build", name)
    os.makedirs(out, exist_ok=True)
    stamp = os.path.join(out, "defines.txt")
    same = os.path.exists(stamp) and open(stamp).read() == defines
    hdr_t = _newest([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] +
                    [os.path.join(os.path.dirname(HERE), "include", "t2s_hip.h")])
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(out, src.replace(".hip", ".o"))
        objs.append(obj)
        if not same or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            jobs.append((src, obj))

    def one(so):
        cmd = [hipcc] + defines.split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, so[0]), "-o", so[1]]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (so[0], r.stderr))
    if jobs:
        with ThreadPoolExecutor(max(1, min(len(jobs), int(os.environ.get("T2S_BUILD_JOBS", "4"))))) as ex:
            list(ex.map(one, jobs))
    lib = os.path.join(out, "libt2s_hip.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    open(stamp, "w").write(defines)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
