"""Build libtranshuman_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m transhuman_amd.build [--force]

The library is built IN-TREE (transhuman_amd/libtranshuman_hip.so) so it
travels with the repository snapshot to the GPU box.  -ffp-contract=off keeps
the mul/add sequences of the sampling, hull and bilinear code un-fused like the
reference's separate torch ops; FMA is used only where written explicitly
(fmaf) or inside MFMA.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libtranshuman_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", f"-I{INC}", f"-I{CSRC}",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (needs ROCm >= 7.0)")
    return exe


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for p in _sources() + headers + [os.path.join(INC, "transhuman_hip.h")]:
        h.update(open(p, "rb").read())
    # (flags without the absolute -I paths: the same tree under another root -- the GPU box's snapshot -- is up to date)
    h.update(" ".join(f for f in FLAGS if not f.startswith("-I")).encode())
    return h.hexdigest()


def needs_build():
    stamp = os.path.join(OBJ, "stamp")
    return not (os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == _digest())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    def cc(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources()))
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(os.path.join(OBJ, "stamp"), "w") as f:
        f.write(_digest())
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) // 1024} KiB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
