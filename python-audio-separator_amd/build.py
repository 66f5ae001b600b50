"""Build libasx.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "asx.hip")
DEPS = [SRC] + sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h")) + \
       [os.path.join(os.path.dirname(HERE), "include", "asx.h")]
OUT = os.path.join(HERE, "libasx.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"]
# The default library carries the shipped kernel of every family plus one fp32-MFMA reference each (the in-library A/B); the superseded generations
# -- conv_wino_kernel, conv_wino2_kernel, conv_winos_kernel, attention_kernel, the ablation instantiations of conv_wino3_kernel / tdf2_kernel /
# tdf3_kernel -- are compiled in only with --experimental (or ASX_EXPERIMENTAL=1): build time and size, profiles/NOTES.md round 6.
if os.environ.get("ASX_EXPERIMENTAL", "0") not in ("", "0") or "--experimental" in sys.argv:
    FLAGS = FLAGS + ["-DASX_EXPERIMENTAL_KERNELS"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


STAMP = OUT + ".srchash"


def source_hash():
    """sha256 over every source the library is built from (names + contents) and the compile flags"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build():
    """True unless libasx.so exists AND was built from exactly the current sources (hash stamp beside it; file times are not
    trusted -- a checkout or a snapshot copy resets them)."""
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    stamp = source_hash()                               # of what is about to be compiled (an edit during the compile must not be stamped as built)
    cmd = [hipcc_path()] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stderr)
    with open(STAMP, "w") as fh:
        fh.write(stamp + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
