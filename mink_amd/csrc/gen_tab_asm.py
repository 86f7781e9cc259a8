"""Generate tab_asm.inc: the QP tableau column of a lane as PINNED VGPRs driven by inline asm.

Why: hipcc cannot keep a 48-double per-lane array in registers through a loop with control flow —
it copies the array between register sets at block boundaries and spills loop-invariant values
into the pivot's critical path (scratch reloads of an LDS address cost ~500 cycles each).  So the
tableau lives in a fixed physical register range and the operations that touch it are emitted here
with literal register numbers:

    zero()                    T[i] = 0
    rank1_prefetch / rank1_body*(lds_addr, g)   T[i] += lds[i]·g   four lane-indexed ds_read_b64 + v_fmac_f64_dpp
                              row_newbcast (row-range variants; *_pub variants also store the next pivot column's entry)
    load_*(lds_addr)          T[i] = lds[i]         whole column / leading rows / residual rows
    get_dyn(k) / set_dyn(k,x) single element, wave-uniform runtime index (VGPR index mode)
    get<I>() / set<I>(x)      single element, compile-time index (tableau build, taps)

Register maps (every primitive takes a `Regs&` first — an empty struct, kept so that call sites do not depend on the map):

Tab<NT> (wave64; TOP = 256 VGPRs per lane at 2 waves/SIMD, 168 at 3 (NT ≤ 24), 128 at 4 (NT ≤ 8)):
    v[TOP-2·NT, TOP)              tableau column (NT doubles): never allocated by the compiler (`amdgpu_num_vgpr` caps it
                                  below the range; asm clobbers make the range count towards the kernel's VGPR total)
    v[TOP-2·NT-S, TOP-2·NT)       the pivot column in 16-lane "planes" (S = 8 registers, see rank1 below)
    v[0, TOP-2·NT-S)              everything the compiler allocates — through the WHOLE kernel, also where the tableau is dead

TabW3<NT> (NT = 32 / 40 / 44): the same with TOP = 168, i.e. 3 waves/SIMD for the 44-row tableau of a humanoid, and S = two
    registers per plane the column really has (6 for 44 rows).  That leaves the compiler 74 registers: enough for the QP
    loops, far too few for forward kinematics / Lie algebra / Jacobian columns (83–236 spilled VGPRs when everything is
    one function) — which therefore run as REAL function calls in these variants (ik_kernel.h pre_phases, wood_start):
    a callee is not bound by the kernel's `amdgpu_num_vgpr` and may use the whole 168-register file, the tableau being
    dead while it runs.  (Tried first and dropped: handing the column to the compiler as pinned "+{v[a:b]}" tuple
    operands of every statement, or of fence statements only, so that it would know where the range is free — hipcc's
    allocator splits and spills 1024-bit pinned tuples at the slightest pressure: 295–2171 spilled VGPRs.)

The staging range must stay ABOVE the compiler's cap even though half of it is only live inside one asm
statement: the VGPRs hipcc reserves for SGPR spills are the highest ones below the cap, reserved
registers are not preserved across an asm that lists them as clobbers (clang warns, and the spilled
SGPRs really are lost — seen as a QP that never converges in the one variant with 400 SGPR spills); and the plane
loads are asynchronous, the compiler must never touch their destination between prefetch and body.
"""

import os

HERE = os.path.dirname(os.path.abspath(__file__))
NTS = (8, 16, 24, 32, 44, 48, 64)
W3_NTS = (32, 44)   # tableaus that also get the 3-waves-per-SIMD map (TabW3<NT>: TOP = 168)
W4_NTS = (16, 24)       # ... and the 4-waves-per-SIMD map (TabW4<NT>: TOP = 128; Tab<16> / Tab<24> are 3-waves maps)
def total_for(nt):
    # VGPRs per lane = 512 / resident waves per SIMD: small tableaus leave room for 4 (NT ≤ 8) or 3 (NT ≤ 24)
    # waves per SIMD instead of 2 — more waves hide more of the serial pivot chain (UR5e-class arms)
    return 128 if nt <= 8 else (168 if nt <= 24 else 256)
NRS = (16, 24, 32, 44, 48)   # leading-row counts of the partial loads / updates (a_stride_for in ik_kernel.h; dof-row prefixes of the low-rank start)
def ntmp_for(nt):
    # Staging registers of the rank-1 update.  Round 1 streamed the pivot column through broadcast ds_read_b128 (two rows
    # per read, 6-8 reads in flight = 24-32 registers).  A broadcast read returns 1 KiB to the wave whatever its address
    # pattern, and at the kernel's residency those returns saturate the CU's LDS path (tools/ubench/gen_rank1_mix.py:
    # 998 ticks per 62-row update against 496 for its FMAs).  Now lane l of every 16-lane row fetches u[16p + l % 16] for
    # the four "planes" p with ONE lane-indexed ds_read_b64 each (4 x 512 B), and the broadcast happens in the DPP operand
    # network: v_fmac_f64_dpp T[i], plane[i / 16], g row_newbcast:(i % 16) — 531 ticks in the same benchmark.
    return 8


def gen(nt: int, total: int = 0, name: str = "Tab") -> str:
    NTMP = ntmp_for(nt)
    TOTAL = total or total_for(nt)
    hi_map = name != "Tab"                   # TabW3 / TabW4: one more resident wave per SIMD than Tab<NT>, callee structure
    if hi_map:
        NTMP = 2 * ((nt + 15) // 16)         # only the planes the column has (44 rows: 3 planes, 6 registers)
    t0 = TOTAL - 2 * nt
    tmp0 = t0 - NTMP
    budget = tmp0
    t1 = t0 + 2 * nt
    treg = lambda i: f"v[{t0 + 2 * i}:{t0 + 2 * i + 1}]"
    clob_tmp = ",".join(f'"v{r}"' for r in range(tmp0, tmp0 + NTMP))
    clob_t = ",".join(f'"v{r}"' for r in range(t0, t1))

    def stmt(body, outs="", ins="", clob="", indent="    "):
        """One asm statement.  outs / ins: operand lists with [names]; clob: further clobbers besides the tableau range."""
        c = ", ".join(x for x in (clob_t, clob) if x)
        return f'{indent}asm volatile("{body}" : {outs} : {ins} : {c});'

    J = "\\n\\t".join
    out = []
    out.append(f"template <> struct {name}<{nt}> {{")
    out.append(f"  static constexpr int kRows = {nt};")
    out.append(f"  static constexpr int kCompilerVgprs = {budget};")
    out.append("  struct Regs {};   // (the column is not a compiler-visible value)")
    # zero
    out.append("  __device__ static __forceinline__ void zero(Regs& t) {")
    out.append(stmt(J(f"v_mov_b32 v{r}, 0" for r in range(t0, t1))))
    out.append("  }")
    # rank1 = prefetch (the four plane loads) + body; the split lets the caller put the reciprocal / multiplier
    # arithmetic between them so that the LDS latency is hidden.
    nplanes = (nt + 15) // 16
    def plane_reg(p):
        return tmp0 + 2 * p
    def fmac(i):
        r = plane_reg(i // 16)
        return f"v_fmac_f64_dpp {treg(i)}, v[{r}:{r + 1}], %[g] row_newbcast:{i % 16} row_mask:0xf bank_mask:0xf"
    def rank1_lines(rows, pub=False):
        """T[i] += u[i]·g for the given rows; the planes were requested by rank1_prefetch.
        pub: the statement first stores %[pub] (this lane's entry of the NEXT pivot column) at LDS address %[pa] and then
        waits for everything older than that store — LDS operations of a wave complete in order, so lgkmcnt(1)
        means "the plane loads are here" without draining the store (look-ahead publishing, ik_kernel.h).
        Otherwise lgkmcnt(0): the compiler may have put LDS / scalar-memory instructions of its own between prefetch and
        body (SMEM returns out of order, so only a full drain is exact); the loads were issued ~100 cycles earlier.
        s_nop 4: a DPP operand must not be read within 5 wait states of an EXEC write by the code before the statement."""
        head = ["ds_write_b64 %[pa], %[pub]", "s_waitcnt lgkmcnt(1)"] if pub else ["s_waitcnt lgkmcnt(0)"]
        return head + ["s_nop 4"] + [fmac(i) for i in rows]
    IN_G = '[g] "v"(g)'
    IN_PUB = '[g] "v"(g), [pa] "v"(pub_addr), [pub] "v"(pub)'
    lines = ["s_waitcnt lgkmcnt(0)"] + [f"ds_read_b64 v[{plane_reg(p)}:{plane_reg(p) + 1}], %[a] offset:{128 * p}" for p in range(nplanes)]
    clob_pre = ",".join(f'"v{r}"' for r in range(tmp0, tmp0 + 2 * nplanes))
    out.append("  // request the pivot column lds[0..kRows): lane l gets lds[16·p + l % 16] in plane p (must be followed by a")
    out.append("  // rank1_body* with the same column)")
    out.append("  __device__ static __forceinline__ void rank1_prefetch(Regs&, unsigned lds_addr) {")
    if hi_map:
        # high-occupancy maps: the lane's plane address is computed inside the statement (mbcnt = lane id), in the first plane's own
        # destination register — as a C++ value it is loop-invariant, lives through the whole QP and, with 72 registers for
        # the compiler, was the value hipcc chose to spill: one scratch reload at the head of every pivot's dependent chain
        pr = plane_reg(0)
        pre = [f"v_mbcnt_lo_u32_b32 v{pr}, -1, 0", f"v_mbcnt_hi_u32_b32 v{pr}, -1, v{pr}", f"v_and_b32 v{pr}, 15, v{pr}",
               f"v_lshl_add_u32 v{pr}, v{pr}, 3, %[a]", "s_waitcnt lgkmcnt(0)"]
        lines2 = pre + [f"ds_read_b64 v[{plane_reg(p)}:{plane_reg(p) + 1}], v{pr} offset:{128 * p}" for p in reversed(range(nplanes))]
        out.append(f'    asm volatile("{J(lines2)}" :: [a] "v"(lds_addr) : {clob_pre}, "memory");')
    else:
        out.append("    const unsigned lane_addr = lds_addr + ((threadIdx.x & 15u) << 3);")
        out.append(f'    asm volatile("{J(lines)}" :: [a] "v"(lane_addr) : {clob_pre}, "memory");')
    out.append("  }")
    out.append("  // T[i] += lds[i]*g; consumes the planes rank1_prefetch requested.")
    out.append("  __device__ static __forceinline__ void rank1_body(Regs& t, unsigned, double g) {")
    out.append(stmt(J(rank1_lines(range(nt))), ins=IN_G, clob=clob_tmp + ', "memory"'))
    out.append("  }")
    out.append("  // rank1_body + look-ahead store of the next pivot column's entry")
    out.append("  __device__ static __forceinline__ void rank1_body_pub(Regs& t, unsigned, double g, unsigned pub_addr, double pub) {")
    out.append(stmt(J(rank1_lines(range(nt), pub=True)), ins=IN_PUB, clob=clob_tmp + ', "memory"'))
    out.append("  }")
    # dynamic row read through the VGPR index mode (uniform runtime index, pinned base register)
    out.append("  // T[k] for a wave-uniform runtime k: s_set_gpr_idx_on + v_mov_b32 relative to the pinned base.")
    out.append("  __device__ static __forceinline__ double get_dyn(Regs& t, int k) {")
    out.append("    int lo, hi;")
    out.append("    const int idx = __builtin_amdgcn_readfirstlane(2 * k);")
    body = f"s_set_gpr_idx_on %[i], gpr_idx(SRC0)\\n\\tv_mov_b32 %[lo], v{t0}\\n\\tv_mov_b32 %[hi], v{t0 + 1}\\n\\ts_set_gpr_idx_off"
    out.append(f'    asm volatile("{body}" : [lo] "=&v"(lo), [hi] "=&v"(hi) : [i] "s"(idx) : "m0");')
    out.append("    return __hiloint2double(hi, lo);")
    out.append("  }")
    # leading rows of a column from LDS (per-lane address): T[i] = lds[i], i < n
    for n in sorted(set([r for r in NRS if r < nt])):
        lines = [f"ds_read_b128 v[{t0 + 4 * k}:{t0 + 4 * k + 3}], %[a] offset:{16 * k}" for k in range(n // 2)]
        lines.append("s_waitcnt lgkmcnt(0)")
        out.append(f"  // T[i] = lds[i], i < {n}")
        out.append(f"  __device__ static __forceinline__ void load_lo_{n}(Regs& t, unsigned lds_addr) {{")
        out.append(stmt(J(lines), ins='[a] "v"(lds_addr)', clob='"memory"'))
        out.append("  }")
    # whole-column load from LDS (per-lane address)
    lines = [f"ds_read_b128 v[{t0 + 4 * k}:{t0 + 4 * k + 3}], %[a] offset:{16 * k}" for k in range(nt // 2)]
    lines.append("s_waitcnt lgkmcnt(0)")
    out.append("  // T[i] = lds[i] for every row (per-lane address: a half-space row's column A[s][:])")
    out.append("  __device__ static __forceinline__ void load_all(Regs& t, unsigned lds_addr) {")
    out.append(stmt(J(lines), ins='[a] "v"(lds_addr)', clob='"memory"'))
    out.append("  }")
    # runtime row write (VGPR index mode on the destination)
    out.append("  // T[k] = x for a wave-uniform runtime k (lanes masked off by exec keep their value).")
    out.append("  __device__ static __forceinline__ void set_dyn(Regs& t, int k, double x) {")
    out.append("    const int lo = __double2loint(x), hi = __double2hiint(x);")
    out.append("    const int idx = __builtin_amdgcn_readfirstlane(2 * k);")
    out.append(stmt(f"s_set_gpr_idx_on %[i], gpr_idx(DST)\\n\\tv_mov_b32 v{t0}, %[lo]\\n\\tv_mov_b32 v{t0 + 1}, %[hi]\\n\\ts_set_gpr_idx_off",
                    ins='[lo] "v"(lo), [hi] "v"(hi), [i] "s"(idx)', clob='"m0"'))
    out.append("  }")
    # partial-row primitives for the low-rank start (rows [0, NR) = the dof rows)
    for nr in [r for r in NRS if r <= nt]:
        out.append(f"  // rank1_body restricted to rows [0, {nr})")
        out.append(f"  __device__ static __forceinline__ void rank1_body_{nr}(Regs& t, unsigned, double g) {{")
        out.append(stmt(J(rank1_lines(range(nr))), ins=IN_G, clob=clob_tmp + ', "memory"'))
        out.append("  }")
    # streamed rank-1 updates (the chain-free dof-block updates of the low-rank start): the statement consumes the planes of
    # THIS row and, as soon as the last FMA that reads a plane has issued, requests that plane of the NEXT row (per-lane
    # address %[an] = next row + 8·(lane % 16)) — the LDS latency of row r + 1 hides under the FMAs of row r.  (An in-flight
    # load only writes its destination long after the FMAs issued before it have read theirs.)
    for nr in sorted(set([r for r in NRS if r < nt] + [nt])):
        lines = ["s_waitcnt lgkmcnt(0)", "s_nop 4"]
        for p in range(nplanes):
            lines += [fmac(i) for i in range(16 * p, min(16 * p + 16, nr))]
            lines.append(f"ds_read_b64 v[{plane_reg(p)}:{plane_reg(p) + 1}], %[an] offset:{128 * p}")
        out.append(f"  // T[i] += lds[i]·g for i < {nr} with the planes already requested; requests the planes at lane address `next`")
        out.append(f"  __device__ static __forceinline__ void rank1_stream_{nr}(Regs& t, unsigned next, double g) {{")
        out.append(stmt(J(lines), ins='[g] "v"(g), [an] "v"(next)', clob=clob_tmp + ', "memory"'))
        out.append("  }")
    # get / set with compile-time index
    out.append("  template <int I> __device__ static __forceinline__ double get(Regs& t) {")
    out.append("    int lo, hi;")
    for i in range(nt):
        kw = "if" if i == 0 else "else if"
        body = f"v_mov_b32 %[lo], v{t0 + 2 * i}\\n\\tv_mov_b32 %[hi], v{t0 + 2 * i + 1}"
        out.append(f'    {kw} constexpr (I == {i}) asm volatile("{body}" : [lo] "=v"(lo), [hi] "=v"(hi));')
    out.append("    else { lo = 0; hi = 0; }")
    out.append("    return __hiloint2double(hi, lo);")
    out.append("  }")
    out.append("  template <int I> __device__ static __forceinline__ void set(Regs& t, double x) {")
    out.append("    const int lo = __double2loint(x), hi = __double2hiint(x);")
    for i in range(nt):
        kw = "if" if i == 0 else "else if"
        body = f"v_mov_b32 v{t0 + 2 * i}, %[lo]\\n\\tv_mov_b32 v{t0 + 2 * i + 1}, %[hi]"
        out.append(f'    {kw} constexpr (I == {i}) asm volatile("{body}" :: [lo] "v"(lo), [hi] "v"(hi) : "v{t0 + 2 * i}", "v{t0 + 2 * i + 1}");')
    out.append("  }")
    out.append("};")
    return "\n".join(out)


KMUS = (18, 24)   # row capacities of the low-rank start's elimination (ik_kernel.h kMu*): task residuals of one problem


def gen_wood_elim() -> str:
    """Low-rank start (ik_kernel.h wood_start): step R of the LDLᵀ elimination of [S | Jh] with one COLUMN per lane in
    compiler-allocated registers z[0..K): z[i] += S[i][R]·g for the rows below R, where the wave-uniform multipliers
    S[i][R] sit in two 16-lane planes (lane l holds entry 16p + l % 16) and are broadcast by the DPP operand network."""
    out = ["template <int R, int K> struct WoodElim;"]
    for K in KMUS:
        for r in range(K):
            rows = list(range(r + 1, K))
            out.append(f"template <> struct WoodElim<{r}, {K}> {{")
            if not rows:
                out.append(f"  __device__ static __forceinline__ void step(double (&)[{K}], double, double, double) {{}}")
            else:
                ops = ", ".join(f'[z{i}] "+v"(z[{i}])' for i in rows)
                body = "\\n\\t".join(["s_nop 4"] + [f"v_fmac_f64_dpp %[z{i}], %[p{i // 16}], %[g] row_newbcast:{i % 16} row_mask:0xf bank_mask:0xf" for i in rows])
                out.append(f"  __device__ static __forceinline__ void step(double (&z)[{K}], double p0, double p1, double g) {{")
                out.append(f'    asm volatile("{body}" : {ops} : [p0] "v"(p0), [p1] "v"(p1), [g] "v"(g));')
                out.append("  }")
            out.append("};")
    # acc[r] += plane[r]·g for every row (the dense Jh·Jhᵀ product of the F_COM builds, ik_kernel.h wood_start)
    out.append("template <int K> struct WoodAll;")
    for K in KMUS:
        ops = ", ".join(f'[z{i}] "+v"(z[{i}])' for i in range(K))
        body = "\\n\\t".join(["s_nop 4"] + [f"v_fmac_f64_dpp %[z{i}], %[p{i // 16}], %[g] row_newbcast:{i % 16} row_mask:0xf bank_mask:0xf" for i in range(K)])
        out.append(f"template <> struct WoodAll<{K}> {{")
        out.append(f"  __device__ static __forceinline__ void step(double (&z)[{K}], double p0, double p1, double g) {{")
        out.append(f'    asm volatile("{body}" : {ops} : [p0] "v"(p0), [p1] "v"(p1), [g] "v"(g));')
        out.append("  }")
        out.append("};")
    return "\n".join(out)


def main():
    parts = ["// GENERATED by gen_tab_asm.py — do not edit.  Pinned-VGPR tableau primitives (see the generator's docstring).",
             "#pragma once", "#include <hip/hip_runtime.h>", "namespace mkh {", "template <int NT> struct Tab;", "template <int NT> struct TabW3;", "template <int NT> struct TabW4;"]
    parts += [gen(nt) for nt in NTS]
    parts += [gen(nt, 168, "TabW3") for nt in W3_NTS]
    parts += [gen(nt, 128, "TabW4") for nt in W4_NTS]
    parts.append(gen_wood_elim())
    parts.append("}  // namespace mkh")
    path, text = os.path.join(HERE, "tab_asm.inc"), "\n".join(parts) + "\n"
    if os.path.exists(path) and open(path).read() == text:
        return                       # (unchanged: keep the timestamp, build.py recompiles what is newer than its object)
    with open(path, "w") as fh:
        fh.write(text)


if __name__ == "__main__":
    main()
