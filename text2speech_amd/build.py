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


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        cmd[1:1] = os.environ.get("T2S_BUILD_DEFINES", "").split()      # e.g. -DT2S_GEMM_ABLATE for the timing-only ablations
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        if verbose:
            sys.stderr.write(r.stderr)
        objs.append(obj)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
