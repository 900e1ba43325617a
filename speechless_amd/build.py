"""Builds speechless_amd/libspeechless_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compile)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PACKAGE_DIR = Path(__file__).resolve().parent
CSRC = PACKAGE_DIR / "csrc"
LIB_PATH = PACKAGE_DIR / "libspeechless_hip.so"
HOST_LIB_PATH = PACKAGE_DIR / "libspeechless_host.so"  # plain C++ helpers of the host input pipeline (no HIP)
HOST_SOURCES = [PACKAGE_DIR / "csrc_host" / "pack_batch.cpp", PACKAGE_DIR / "csrc_host" / "beam_search.cpp"]
CXX = os.environ.get("CXX", "g++")
SOURCES = ["capi.hip", "conv_nt_bf16.hip", "wgrad_tn_bf16.hip", "conv_f32.hip", "ctc.hip", "misc.hip", "spectrogram.hip", "conv_chain_bf16.hip",
           "conv1x1_bwd_bf16.hip", "split3.hip"]
# translation units built a SECOND time from the same source with -DSL_ELEM_F16: the NT / TN kernels on v_mfma_*_f16 for the
# f16x3 parity path (csrc/common.h: SL_MFMA16; only the fp32 / plane-output instantiations, about a third of the bf16 build)
F16_VARIANTS = {"conv_nt_f16": "conv_nt_bf16.hip", "wgrad_tn_f16": "wgrad_tn_bf16.hip"}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("SL_EXTRA_FLAGS", "").split()  # experiments only (e.g. -DSL_NT_SETPRIO); the default build has none


def _newest_source_mtime():
    files = [CSRC / s for s in SOURCES] + [CSRC / "common.h", PACKAGE_DIR.parent / "include" / "speechless_hip.h"]
    return max(f.stat().st_mtime for f in files)


# These files read LDS fragments through inline asm with hand-counted s_waitcnt (the compiler does not know the
# registers are still being filled).  A register spill would store such a register before its data has arrived, so a
# kernel of these files that needs scratch memory is a BUILD ERROR, not a slow kernel.
NO_SCRATCH = {"conv_nt_bf16.hip", "wgrad_tn_bf16.hip", "conv_chain_bf16.hip", "conv1x1_bwd_bf16.hip"}


def _scratch_users(remarks):
    """kernel names with ScratchSize > 0 in hipcc -Rpass-analysis=kernel-resource-usage output"""
    bad, name = [], None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[-Rpass")[0].strip()
        elif "ScratchSize [bytes/lane]:" in line:
            size = int(line.split("ScratchSize [bytes/lane]:")[1].split("[-Rpass")[0].strip())
            if size > 0:
                bad.append("{} ({} bytes/lane)".format(name, size))
    return bad


# ctc.hip: the probability-domain lattice is a lone wave's dependent chain of scalar fp32 operations; packed fp32 VALU
# (v_pk_mul_f32 out of the SLP vectorizer) costs such a wave more than the two scalar instructions it replaces
FILE_FLAGS = {"ctc.hip": ["-fno-slp-vectorize"]}


def _compile(unit):
    """unit: a source file name, or a key of F16_VARIANTS (the same source compiled with -DSL_ELEM_F16 into <key>.o)"""
    src = F16_VARIANTS.get(unit, unit)
    obj = CSRC / ((unit + ".o") if unit in F16_VARIANTS else src.replace(".hip", ".o"))
    extra = ["-DSL_ELEM_F16"] if unit in F16_VARIANTS else []
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + extra + ["-c", str(CSRC / src), "-o", str(obj)]
    if src in NO_SCRATCH:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for {}:\n{}\n{}".format(src, res.stdout, res.stderr))
    err = res.stderr
    if src in NO_SCRATCH:
        bad = _scratch_users(err)
        if bad and os.environ.get("SL_ALLOW_SCRATCH"):  # experiments only: results of the listed kernels are not to be trusted
            print("WARNING: scratch in asm-read kernels: " + "; ".join(b.split("(")[0][-60:] for b in bad[:4]), file=sys.stderr)
            bad = []
        if bad:
            raise RuntimeError("{}: kernels with asm-tracked LDS reads must not spill, but these use scratch: {}".format(
                src, "; ".join(bad)))
        err = "\n".join(l for l in err.splitlines() if "-Rpass-analysis" not in l)
    return obj, err


def build_host(force=False):
    newest = max(f.stat().st_mtime for f in HOST_SOURCES + [PACKAGE_DIR.parent / "include" / "speechless_host.h"])
    if not force and HOST_LIB_PATH.exists() and HOST_LIB_PATH.stat().st_mtime >= newest:
        return HOST_LIB_PATH
    cmd = [CXX, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-o", str(HOST_LIB_PATH)] + \
        [str(f) for f in HOST_SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("host library build failed:\n{}\n{}".format(res.stdout, res.stderr))
    return HOST_LIB_PATH


def build(force=False, verbose=False):
    build_host(force)
    if not force and LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= _newest_source_mtime():
        return LIB_PATH
    units = SOURCES + list(F16_VARIANTS)
    # (the two big files first: each of their translation units takes minutes, the rest seconds)
    units.sort(key=lambda u: 0 if F16_VARIANTS.get(u, u) in ("conv_nt_bf16.hip", "wgrad_tn_bf16.hip") else 1)
    with ThreadPoolExecutor(max_workers=len(units)) as pool:
        results = list(pool.map(_compile, units))
    objs = [str(o) for o, _ in results]
    if verbose:
        for (_, err), src in zip(results, units):
            if err.strip():
                print("[{}]\n{}".format(src, err), file=sys.stderr)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n{}\n{}".format(res.stdout, res.stderr))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
