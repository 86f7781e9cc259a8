"""Build libminkhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mink_amd.csrc.build            # or: python mink_amd/csrc/build.py [--force]

Each kernel variant ik_solve_kernel<NT, FEAT> is compiled in its own generated translation unit
(_build/variant_<NT>_<FEAT>.hip) so that the variants build in parallel; minkhip.hip holds the host
side of the C ABI.  The shared library is written next to the Python package
(mink_amd/libminkhip.so) so that it travels with the repo snapshot; it is git-ignored (*.so).
"""

import concurrent.futures
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
# Same-box A/B of compiler flags: MKH_BUILD_TAG=x builds _build_x/ → libminkhip_x.so next to the default library
# (mink_amd/_native.py loads it when MKH_LIB_TAG=x); unset = the product build.
_TAG = os.environ.get("MKH_BUILD_TAG", "")
LIB = os.path.join(PKG, f"libminkhip_{_TAG}.so" if _TAG else "libminkhip.so")
RESOURCES = os.path.join(PKG, f"kernel_resources_{_TAG}.json" if _TAG else "kernel_resources.json")   # written next to the library at build time (_write_resources)
BUILD = os.path.join(HERE, f"_build_{_TAG}" if _TAG else "_build")
def _sources() -> list:
    """Every file the library is built from: whatever sits in csrc/ (kernel headers, generators, host code) plus the
    public header — derived from the directory, not listed by hand (a hand-kept list once missed convex_dev.h)."""
    fs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE))
          if f.endswith((".h", ".hip", ".py", ".inc")) and f not in ("tab_asm.inc", "__init__.py")]
    return fs + [os.path.join(HERE, "..", "..", "include", "minkhip.h")]


NTS = (8, 16, 24, 32, 44, 48, 64)          # tableau rows per lane (must match kVariants in minkhip.hip)
FEATS = (0, 16, 6, 8, 72, 88, 136, 30, 31, 256)      # lean, + fused step loop, RelativeFrameTask/ComTask, collision rows only, collision rows with plane / sphere / capsule pairs only, the same + fused step loop, collision rows with general convex pairs (GJK), every feature but taps, everything, dense (plugin) rows only (ik_kernel.h F_*)
# low-rank ("Woodbury") start: (NT, NR) = (tableau rows ≥ nv + task rows, dof rows ≥ nv); F_WOOD = 32
WOOD = ((16, 16), (24, 24), (32, 32), (44, 44), (48, 48))   # (NT, NR): the task residuals are eliminated outside the tableau (ik_kernel.h wood_start), NT = NR ≥ nv; must match kWoodVariants in minkhip.hip
# 3-waves-per-SIMD register map (TabW3 in tab_asm.inc, 168 VGPRs, compact LDS layout): (NT, FEAT) without collision rows
W3 = ((44, 0),)                        # must match kW3Variants in minkhip.hip
W3_WOOD = ((44, 44, 32), (44, 44, 48), (32, 32, 32), (32, 32, 48),    # (NT, NR, FEAT) of the low-rank start with the 3-waves map
           (24, 24, 32), (24, 24, 48), (16, 16, 32), (16, 16, 48),    # ... and, for NT ≤ 24, the 4-waves map (TabW4; same suffix _w3: "one more wave")
           (44, 44, 36), (44, 44, 52))                                # F_COM (ComTask rows / up to 24 task rows) and its fused loops: round 5, ik_kernel.h MKH_WOOD_SPLIT
# one problem per workgroup (round 6; ik_kernel.h MKH_ONE_SHOT): a twin of the humanoid-size one-more-wave build WITHOUT the persistent
# loop's machinery (ticket draws, double-buffered inputs, loop-carried scalars) — launched when the grid is the batch (minkhip.hip launch()).
# Measured: `44_32_r44_w3o` 0.713 -> 0.703 ms on the headline; the F_COM twin `44_36_r44_w3o` (12 spilled VGPRs) 0.350 -> 0.352 ms at
# 16 384 instances of the G1 full example — not built.
W3_WOOD_ONE_SHOT = ((44, 44, 32),)
WOOD_FEATS = (32, 48, 33, 36, 52)      # F_WOOD, | F_STEPS, | F_TAPS (cycle-counter profiling only), | F_COM, | F_COM | F_STEPS
# low-rank start WITH half-space rows (round 6; ik_kernel.h wood_start "half-space rows"): (NT, NR, FEAT) on the 2-waves map,
# NR = NT (the rows' columns of the elimination sit behind the dofs in the tableau-wide rows of Jh); must match kWoodRowVariants
# in minkhip.hip.  40 = F_WOOD | F_COLL (analytic pairs, phases as calls)
WOOD_ROWS = ((48, 48, 40),)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-null-conversion"]
# experiments: extra compiler flags for every translation unit (e.g. MKH_EXTRA_FLAGS="-DMKH_FORCE_COLL_CALL"); part of the
# per-object command tag, so changing it recompiles
FLAGS += os.environ.get("MKH_EXTRA_FLAGS", "").split()
# Kernel TUs only: MachineLICM hoists every constant materialisation (v_mov of fp64 polynomial
# coefficients) and lane-address computation out of the per-problem loop, where it then lives across
# the whole QP and gets SPILLED (G1 lean variant: 31 VGPR + 31 SGPR spills, 128 B scratch per lane whose
# write-back was 30 % of the kernel's HBM writes).  Without the pass the same kernel has no spills.
KERNEL_FLAGS = ["-mllvm", "-disable-machine-licm",
                # the pinned tableau / staging registers sit above the amdgpu_num_vgpr cap on purpose: clang reports
                # every asm clobber of them as "reserved" (≈2 000 warnings per build; tools/check_vgpr_cap.py is the check)
                "-Wno-inline-asm"]


def _hipcc() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES) or not os.path.exists(BUILD_INFO):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources()) \
        or not os.path.exists(os.path.join(HERE, "tab_asm.inc"))      # (generated by gen_tab_asm.py, untracked)


def _deps(depfile: str) -> list:
    """Prerequisites from a compiler-written Makefile fragment (-MD -MF)."""
    txt = open(depfile).read().replace("\\\n", " ")
    return [d for d in txt.split(":", 1)[1].split() if d]


def _obj_stale(src: str, obj: str, cmd_tag: str) -> bool:
    """A translation unit is recompiled when its object is missing, was built with other flags, or is older than the
    source or any header the compiler reported for it (so an incremental rebuild compiles only what an edit touches)."""
    dep, tag = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(tag)) or open(tag).read() != cmd_tag:
        return True
    t = os.path.getmtime(obj)
    try:
        return any(os.path.getmtime(d) > t for d in [src] + _deps(dep))
    except OSError:
        return True


def _parse_resource_remarks(stderr: str) -> dict:
    """{function: {vgprs, sgpr_spills, vgpr_spills, scratch_bytes_per_lane, occupancy_waves_per_simd}} from the compiler's
    -Rpass-analysis=kernel-resource-usage remarks of one translation unit (kernels and their real callees)."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}
    for m in re.finditer(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", stderr):
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    return out


def _short_name(mangled: str) -> str:
    m = re.match(r"_ZN(?:3mkh|12_GLOBAL__N_1)(\d+)", mangled)
    if not m:
        return mangled
    n, a = int(m.group(1)), m.end()
    name, rest = mangled[a:a + n], mangled[a + n:]
    if rest.startswith("I"):                      # template arguments: integer / bool literals (lane kernels)
        args = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if args:
            name += "<" + ",".join(re.findall(r"L[ib](\d+)E", args.group(1))) + ">"
    return name


def _sha256(path: str) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _device_code_sha256(obj: str) -> str:
    """SHA-256 of the DEVICE code of a translation unit: the .text and .rodata (kernel descriptors) sections of the gfx950 code
    object inside the host object's fat binary.  The host object itself carries the build directory (the module id of the fat
    binary wrapper is derived from the source path), so its hash changed with the directory the library was built in and every
    committed profile looked stale after a rebuild elsewhere (round-4 review); the device code does not."""
    import hashlib
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    try:
        with tempfile.TemporaryDirectory() as d:
            fat, co = os.path.join(d, "x.fatbin"), os.path.join(d, "x.co")
            subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(d, "x.o")], check=True, capture_output=True)
            subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
            h = hashlib.sha256()
            for sec in (".text", ".rodata"):
                out = os.path.join(d, "sec.bin")
                subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", f"--only-section={sec}", co, out], check=True, capture_output=True)
                with open(out, "rb") as fh:
                    h.update(fh.read())
            return h.hexdigest()
    except (OSError, subprocess.CalledProcessError):
        return _sha256(obj)                # (no LLVM tools: fall back to the object file)


def _git_state() -> dict:
    """HEAD and whether the sources under it were modified when the library was linked (the GPU box has no .git)."""
    root = os.path.dirname(PKG)
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "mink_amd", "include"], capture_output=True,
                               text=True, check=True).stdout.strip() != ""
        return {"git_head": head, "git_dirty_sources": dirty}
    except (OSError, subprocess.CalledProcessError):
        return {"git_head": None, "git_dirty_sources": None}


BUILD_INFO = os.path.join(PKG, f"build_info_{_TAG}.json" if _TAG else "build_info.json")


def _write_resources(objs: list) -> None:
    """mink_amd/kernel_resources.json: registers, spills and scratch of every kernel in the library, from the compiler's
    own remarks at build time (bench.py reports `vgpr_spills` of the kernel it ran from here; tests/test_abi.py reads it)."""
    table = {}
    for o in objs:
        path = o + ".res.json"
        if not os.path.exists(path):
            continue
        fns = json.load(open(path))
        kernels = {f: r for f, r in fns.items() if "ik_solve_kernel" in f or "ik_lane_kernel" in f or "_kernel" in _short_name(f)}
        callees = {f: r for f, r in fns.items() if f not in kernels}
        code = _device_code_sha256(o)      # the translation unit's DEVICE code: what the kernel's counters are a property of
        for f, r in kernels.items():
            e = dict(r)
            e["code_sha256"] = code
            # real callees of the 3-waves builds (pre_phases, wood_start, ...) run inside the kernel: their spills count
            e["callees"] = {_short_name(c): cr for c, cr in callees.items()}
            e["vgpr_spills_with_callees"] = r.get("vgpr_spills", 0) + sum(cr.get("vgpr_spills", 0) for cr in callees.values())
            e["sgpr_spills_with_callees"] = r.get("sgpr_spills", 0) + sum(cr.get("sgpr_spills", 0) for cr in callees.values())
            table[_short_name(f)] = e
    with open(RESOURCES, "w") as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    # provenance of the library itself (tools/rocprof_summary.py stamps profiles with it, bench.py compares)
    info = {"library": os.path.basename(LIB), "library_sha256": _sha256(LIB)}
    info.update(_git_state())
    with open(BUILD_INFO, "w") as fh:
        json.dump(info, fh, indent=1, sort_keys=True)


def _write_if_changed(path: str, text: str) -> None:
    if os.path.exists(path) and open(path).read() == text:
        return
    with open(path, "w") as fh:
        fh.write(text)


def _generate() -> list:
    os.makedirs(BUILD, exist_ok=True)
    subprocess.run([sys.executable, os.path.join(HERE, "gen_tab_asm.py")], check=True)
    srcs = []
    for nt in NTS:
        for ft in FEATS:
            name = f"variant_{nt}_{ft}"
            _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_FEAT {ft}
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
            srcs.append(name)
    for nt, nr in WOOD:
        for ft in WOOD_FEATS:
            name = f"variant_{nt}_{ft}_r{nr}"
            _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_NR {nr}
#define MKH_FEAT {ft}
#define MKH_KERNEL_NAME ik_solve_kernel_{nt}_{ft}_r{nr}
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}_r{nr}(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}_r{nr}, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
            srcs.append(name)
    for nt, nr, ft in WOOD_ROWS:
        name = f"variant_{nt}_{ft}_r{nr}"
        _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_NR {nr}
#define MKH_FEAT {ft}
#define MKH_KERNEL_NAME ik_solve_kernel_{nt}_{ft}_r{nr}
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}_r{nr}(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}_r{nr}, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
        srcs.append(name)
    for nt, ft in W3:
        name = f"variant_{nt}_{ft}_w3"
        _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_FEAT {ft}
#define MKH_W3 1
#define MKH_KERNEL_NAME ik_solve_kernel_{nt}_{ft}_w3
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}_w3(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}_w3, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
        srcs.append(name)
    for nt, nr, ft in W3_WOOD:
        name = f"variant_{nt}_{ft}_r{nr}_w3"
        _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_NR {nr}
#define MKH_FEAT {ft}
#define MKH_W3 1
{"#define MKH_W4 1" if nt <= 24 else ""}
#define MKH_KERNEL_NAME ik_solve_kernel_{nt}_{ft}_r{nr}_w3
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}_r{nr}_w3(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}_r{nr}_w3, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
        srcs.append(name)
    for nt, nr, ft in W3_WOOD_ONE_SHOT:
        name = f"variant_{nt}_{ft}_r{nr}_w3o"
        _write_if_changed(os.path.join(BUILD, name + ".hip"), f"""// generated by build.py
#define MKH_NT {nt}
#define MKH_NR {nr}
#define MKH_FEAT {ft}
#define MKH_W3 1
#define MKH_ONE_SHOT 1
#define MKH_KERNEL_NAME ik_solve_kernel_{nt}_{ft}_r{nr}_w3o
#include "../ik_kernel.h"
namespace mkh {{
void launch_{nt}_{ft}_r{nr}_w3o(int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P, const SolveArgs& a,
                      const TapArgs* taps) {{
  hipLaunchKernelGGL(ik_solve_kernel_{nt}_{ft}_r{nr}_w3o, dim3(grid), dim3(kWave), lds_bytes, stream, P, a, taps);
}}
}}  // namespace mkh
""")
        srcs.append(name)
    decls = "\n".join(f"void launch_{nt}_{ft}(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                      for nt in NTS for ft in FEATS)
    decls += "\n" + "\n".join(f"void launch_{nt}_{ft}_r{nr}_w3o(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                             for nt, nr, ft in W3_WOOD_ONE_SHOT)
    decls += "\n" + "\n".join(f"void launch_{nt}_{ft}_r{nr}(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                             for nt, nr in WOOD for ft in WOOD_FEATS)
    decls += "\n" + "\n".join(f"void launch_{nt}_{ft}_r{nr}(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                             for nt, nr, ft in WOOD_ROWS)
    decls += "\n" + "\n".join(f"void launch_{nt}_{ft}_w3(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                             for nt, ft in W3)
    decls += "\n" + "\n".join(f"void launch_{nt}_{ft}_r{nr}_w3(int, int, hipStream_t, const DeviceProblem*, const SolveArgs&, const TapArgs*);"
                             for nt, nr, ft in W3_WOOD)
    cases0 = "\n".join(f"  if (one_shot && w3 && nt == {nt} && nr == {nr} && feat == {ft}) {{ launch_{nt}_{ft}_r{nr}_w3o(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                       for nt, nr, ft in W3_WOOD_ONE_SHOT)
    cases0 += "\n  if (one_shot) return -1;\n"
    cases0 += "\n".join(f"  if (w3 && nt == {nt} && nr == {nr} && feat == {ft}) {{ launch_{nt}_{ft}_r{nr}_w3(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                       for nt, nr, ft in W3_WOOD)
    cases = cases0 + "\n" + "\n".join(f"  if (w3 && nt == {nt} && nr == 0 && feat == {ft}) {{ launch_{nt}_{ft}_w3(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                      for nt, ft in W3)
    cases += "\n" + "\n".join(f"  if (nt == {nt} && nr == 0 && feat == {ft}) {{ launch_{nt}_{ft}(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                      for nt in NTS for ft in FEATS)
    cases += "\n" + "\n".join(f"  if (nt == {nt} && nr == {nr} && feat == {ft}) {{ launch_{nt}_{ft}_r{nr}(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                              for nt, nr in WOOD for ft in WOOD_FEATS)
    cases += "\n" + "\n".join(f"  if (nt == {nt} && nr == {nr} && feat == {ft}) {{ launch_{nt}_{ft}_r{nr}(grid, lds_bytes, stream, P, a, taps); return 0; }}"
                              for nt, nr, ft in WOOD_ROWS)
    _write_if_changed(os.path.join(BUILD, "dispatch.hip"), f"""// generated by build.py
#include <hip/hip_runtime.h>
#include "../mkh_types.h"
namespace mkh {{
{decls}
int launch_variant(int nt, int nr, int feat, bool w3, int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P,
                   const SolveArgs& a, const TapArgs* taps, bool one_shot) {{
{cases}
  return -1;
}}
}}  // namespace mkh
""")
    srcs.append("dispatch")
    # translation units of variants that are no longer built (tools/check_vgpr_cap.py walks the directory)
    for f in os.listdir(BUILD):
        if f.startswith("variant_") and f.endswith((".hip", ".o")) and f.rsplit(".", 1)[0] not in srcs:
            os.remove(os.path.join(BUILD, f))
    return srcs


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    srcs = _generate()
    jobs = [(os.path.join(BUILD, s + ".hip"), os.path.join(BUILD, s + ".o")) for s in srcs]
    jobs.append((os.path.join(HERE, "minkhip.hip"), os.path.join(BUILD, "minkhip.o")))
    jobs.append((os.path.join(HERE, "lane_variants.hip"), os.path.join(BUILD, "lane_variants.o")))
    jobs.append((os.path.join(HERE, "wide_variants.hip"), os.path.join(BUILD, "wide_variants.o")))
    jobs.append((os.path.join(HERE, "convex_pre.hip"), os.path.join(BUILD, "convex_pre.o")))

    def compile_one(job):
        src, obj = job
        extra = KERNEL_FLAGS if os.path.basename(src).startswith("variant_") else []
        cmd = [hipcc] + FLAGS + extra
        tag = " ".join(cmd)
        if not force and not _obj_stale(src, obj, tag):
            return obj, False
        r = subprocess.run(cmd + ["-MD", "-MF", obj + ".d", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj],
                           stdout=subprocess.DEVNULL if not verbose else None, stderr=subprocess.PIPE, text=True)
        rest = [ln for ln in r.stderr.split("\n") if "kernel-resource-usage" not in ln]
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(obj + ".res.json", "w") as fh:
            json.dump(_parse_resource_remarks(r.stderr), fh)
        with open(obj + ".cmd", "w") as fh:
            fh.write(tag)
        del rest
        return obj, True

    if verbose:
        print(f"[mink_amd] {len(jobs)} translation units for gfx950 (recompiling what changed) ...", flush=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        res = list(ex.map(compile_one, jobs))
    objs = [o for o, _ in res]
    if verbose:
        print(f"[mink_amd] compiled {sum(1 for _, c in res if c)} of {len(jobs)}", flush=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    _write_resources(objs)
    if verbose:
        print(f"[mink_amd] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
