"""In-tree build of libwfpt.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU. The .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libwfpt.so")
SOURCES = ["wfpt_kernels.hip", "wfpt_api.hip", "wfpt_bvh_build.hip", "wfpt_host.cpp"]
HEADERS = [os.path.join(CSRC, "wfpt_kernels.h"), os.path.join(CSRC, "wfpt_bvh4.h"), os.path.join(CSRC, "wfpt_device_math.h"),
           os.path.join(ROOT, "include", "wfpt.h")]
# -ffp-contract=off: results must be bit-identical to the oracle, the only fused ops are explicit fmaf.
# Correctly rounded fp32 divide/sqrt is hipcc's default and is requested explicitly anyway.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libwfpt.so cannot be built (and there is no CPU fallback)")
    return exe


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


BUILD_INFO = os.path.join(PKG_DIR, "build_info.json")


def source_digest():
    """sha256 over the sources libwfpt.so is built from (what a measurement was taken on, wherever git is not at hand)."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def write_build_info():
    """Provenance for bench lines and profiles: the GPU box gets a snapshot without .git, so the commit (and whether the tree was
    dirty) is stamped here, next to the library, when it is built. Git-ignored like the library; travels with it."""
    import datetime
    import json

    def git(*a):
        try:
            return subprocess.run(["git", "-C", ROOT] + list(a), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout.strip()
        except Exception:
            return ""
    info = {"git_head": git("rev-parse", "HEAD") or None, "git_dirty": bool(git("status", "--porcelain", "--untracked-files=no")),
            "built_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "source_sha256": source_digest(),
            "flags": FLAGS + os.environ.get("WFPT_EXTRA_FLAGS", "").split()}
    try:
        json.dump(info, open(BUILD_INFO, "w"), indent=1)
    except OSError:
        pass
    return info


def build_info():
    """What write_build_info() left beside the library, checked against the sources as they are now."""
    import json
    try:
        info = json.load(open(BUILD_INFO))
    except Exception:
        info = {"git_head": None}
    try:
        info["sources_match_build"] = info.get("source_sha256") == source_digest()
    except OSError:
        info["sources_match_build"] = None
    return info


def build(force=False, verbose=False):
    """Compile every HIP source into wavefront_path_tracer_amd/libwfpt.so; returns the path."""
    if not force and not stale():
        return LIB_PATH
    out = os.environ.get("WFPT_LIB_OUT", LIB_PATH)  # tuning builds: WFPT_EXTRA_FLAGS="-D..." WFPT_LIB_OUT=...
    cmd = [hipcc()] + FLAGS + os.environ.get("WFPT_EXTRA_FLAGS", "").split()
    cmd += ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout)
    if out == LIB_PATH:
        write_build_info()
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
