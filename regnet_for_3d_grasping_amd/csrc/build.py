"""Builds libregnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One shared object, in-tree (regnet_for_3d_grasping_amd/csrc/libregnet_hip.so) so it travels to the
GPU box with the repo snapshot.  geometry.hip is compiled with -ffp-contract=off: the indices it
emits depend on individually rounded fp32 distance arithmetic.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libregnet_hip.so")
ARCH = "gfx950"

# (source, extra flags)
SOURCES = [
    ("api.hip", []),
    ("geometry.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),  # packed f32 VALU slows the FPS scan (measured -7%)
    ("gather.hip", []),
    ("region.hip", ["-ffp-contract=off"]),
    ("grid.hip", ["-ffp-contract=off"]),
    ("mlp.hip", []),   # (mlp_gemm_kernel<0>'s unsatisfiable occupancy request is silenced by a pragma around that kernel only)
    ("sa_chain.hip", ["-fno-slp-vectorize"]),   # SLP turns the layer-1 FMAs into v_pk_mul + separate adds
    ("rowchain.hip", ["-fno-slp-vectorize"]),
    ("heads.hip", []),
    ("heads_train.hip", []),
    ("losses.hip", ["-ffp-contract=off"]),
    ("tgemm.hip", []),
    ("sa_split.hip", ["-fno-slp-vectorize"]),   # experiment, off by default: the level-1 chain on the bf16 pipe (fused.SPLIT_PRODUCTS)
    ("tsplit.hip", []),      # experiment, off by default: split products on the bf16 matrix pipe (conv1x1_train.SPLIT_PRODUCTS)
    ("bn_train.hip", []),
    ("np_random.hip", []),
    ("np_random_dev.hip", []),
    ("dataset.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stamp():
    h = hashlib.sha256()
    for name in sorted(os.listdir(HERE)):
        if name.endswith((".hip", ".h", ".py")):
            with open(os.path.join(HERE, name), "rb") as f:
                h.update(name.encode()); h.update(f.read())
    inc = os.path.join(HERE, "..", "..", "include", "regnet_hip.h")
    with open(inc, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


REPO = os.path.dirname(os.path.dirname(HERE))
# kernels whose register budget is part of their design: a build in which one of them spills vector registers to scratch is an
# error, not a slow kernel found later in a profile (object file -> mangled-name fragments)
NO_VGPR_SPILL = {"mlp.o": ["mlp_gemm_kernelILi0E", "gemm2_kernel"], "sa_chain.o": ["sa_chain_kernel"],
                 "heads.o": ["heads_chain_kernel", "heads_tree_kernel"]}


def check_no_vgpr_spill(obj_path, fragments):
    """Read the gfx950 code object's kernel metadata out of a host object's fat binary and raise if a kernel whose mangled name
    contains one of ``fragments`` has a non-zero ``.vgpr_spill_count``.  Skipped (returns None) when the LLVM tools are absent."""
    import shutil
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(llvm, "llvm-objdump"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        return None
    tmp = tempfile.mkdtemp(prefix="regnet_co_")
    try:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([objdump, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found = 0
        for name in os.listdir(tmp):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, name)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            kernel = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith(".name:"):
                    kernel = line.split(":", 1)[1].strip()
                elif line.startswith(".vgpr_spill_count:") and kernel and any(f in kernel for f in fragments):
                    found += 1
                    if int(line.split(":", 1)[1]) != 0:
                        raise RuntimeError("%s: %s spills %s vector registers" % (os.path.basename(obj_path), kernel, line.split(":", 1)[1].strip()))
        return found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
# measurement twins of product sources: the product translation units carry no measurement branches; a twin is GENERATED from
# the product source by a committed patch (scripts/ablate/*.patch) at build time, so it cannot drift behind the product --
# a patch that no longer applies fails the measurement build instead of measuring another kernel
MEASURE_PATCHES = {"geometry.hip": os.path.join(REPO, "scripts", "ablate", "geometry_measure.patch")}


def measurement_twin(src, out_dir):
    """Apply the measurement patch of product source ``src`` (a name in MEASURE_PATCHES) and return the generated file's path."""
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, src.replace(".hip", "_measure.hip"))
    with open(MEASURE_PATCHES[src], "rb") as f:
        patch = f.read()
    res = subprocess.run(["patch", "-s", "-o", out, os.path.join(HERE, src)], input=patch, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("%s no longer applies to csrc/%s (re-base the measurement branches):\n%s" % (
            MEASURE_PATCHES[src], src, res.stdout.decode(errors="replace")))
    return out
CHAIN_TRACE = ["-DCH_TRACE_H=\"%s\"" % os.path.join(REPO, "scripts", "ablate", "chain_trace.h")]


def build_variant(out_path, defines, measure=False):
    """An extra copy of the library with -D defines applied to every source (A/B measurements only, e.g.
    ``build_variant('/tmp/x.so', ['-DRC_WAVES=4'])``); the product library is ``build()``'s.  ``measure=True`` swaps in
    the measurement twins generated from ``MEASURE_PATCHES`` (the FPS_ABLATE / FPS_ONE_BARRIER / FPS_FORCE_MULTI branches)."""
    hipcc = _hipcc()
    objs = []
    tmp = out_path + ".objs"
    os.makedirs(tmp, exist_ok=True)
    procs = []
    for src, extra in SOURCES:
        obj = os.path.join(tmp, src.replace(".hip", ".o"))
        path = measurement_twin(src, tmp) if (measure and src in MEASURE_PATCHES) else os.path.join(HERE, src)
        procs.append(subprocess.Popen([hipcc] + COMMON + extra + ["-I", HERE] + list(defines) + ["-c", path, "-o", obj]))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out_path] + objs)
    return out_path


def build(force=False, verbose=False):
    stamp_file = os.path.join(HERE, ".build_stamp")
    stamp = _stamp()
    if not force and os.path.exists(OUT) and os.path.exists(stamp_file):
        with open(stamp_file) as f:
            if f.read().strip() == stamp:
                return OUT
    hipcc = _hipcc()
    objs = []
    procs = []
    for src, extra in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [hipcc] + COMMON + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out:
            print(out.decode())
    for obj_name, fragments in NO_VGPR_SPILL.items():
        obj = os.path.join(HERE, obj_name)
        if os.path.exists(obj):
            check_no_vgpr_spill(obj, fragments)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
