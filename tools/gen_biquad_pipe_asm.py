#!/usr/bin/env python3
"""Writes graphaudio_amd/csrc/ga_biquad_pipe_asm.inc: the steady-state walk of biquad_pipe_kernel (ga_kernels.hip) as ONE inline
assembly statement -- a run of `nb` batches of 16 pipeline steps, 8 vector instructions per step, fixed registers.

Why assembly: a wave that is alone on its SIMD pays ~5 cycles per instruction whatever the instruction is (measured,
tools/micro/bq_pipe_probe.hip: 221 instructions per batch = 1131 cycles), so the walk costs its instruction COUNT.  The compiler's
version of the same order needs 13.8 instructions per step (hazard s_nops between separate asm statements -- it has to assume
every asm result is a partial-register write --, moves, a DPP move plus a select); this one needs 8 + 1 per step for the LDS
traffic and the loop.

One step of section lane q (sample i of the lane; BiQuadFilterNode.cs:137-138, every operation rounded separately, in the
reference's order):
    a  A[i]  = {a1, b1} * w[i-1]                 v_pk_mul_f32 (both halves take the low word of w's pair)
    b  m     = b0 * w[i-1]                        (the output half runs one step behind ...)
    c  t     = x[i] - A[i].lo
    d  B[i]  = {a2, b2} * w[i-1]
    e  w[i]  = t - B[i-1].lo
    f  y[i-2] = s[i-2] + B[i-3].hi                (... and its last addition two)
    g  s[i-1] = m + A[i-1].hi
    h  x[i+1] = section 0 ? the cascade's sample i+1 : y[i-3] of the lane to the left      v_cndmask_b32_dpp row_shr:1, vcc = section-0 lanes
No instruction follows its producer (a dependent result would cost ~4 more cycles), the DPP source is 8 instructions old.
The loop body is rotated by two steps (2..15, 0', 1') so that a batch's last outputs are complete when it ends.

Entered from a CLEAN state (w[-1], w[-2], w[-3] and the complete y[-1..-4]): y[-1] is recomputed from the real operands,
y[-2] is reproduced as y[-2] + (-0.0).  Left in a clean state (epilogue).
"""
import os

XV = list(range(192, 208))          # the cascade's samples of this batch (section-0 lanes), refreshed in place for the next batch
YV = list(range(208, 224))          # y of steps 0..15
WP = [(224, 225), (226, 227)]       # w[i] in WP[i % 2].lo
AP = [(228, 229), (230, 231)]
BP = [(232, 233), (234, 235), (236, 237), (238, 239)]
T, M = 240, 241
S = [242, 243]
X = [244, 245]
AIN, AOUT, AINC, W3 = 246, 247, 248, 249
TP = (250, 251)                     # w[-3] on entry
CLOBBER = list(range(192, 252))


def pair(p):
    return f"v[{p[0]}:{p[1]}]"


def step(i, out, first_use_wait=False, parts="abcdefgh"):
    """uniform mid-stream step i (any integer; registers rotate)"""
    a_cur, a_prev = AP[i % 2], AP[(i + 1) % 2]
    w_prev, w_cur = WP[(i + 1) % 2], WP[i % 2]
    if "a" in parts:
        out.append(f"v_pk_mul_f32 {pair(a_cur)}, %[ab1], {pair(w_prev)} op_sel_hi:[1,0]")
    if "b" in parts:
        out.append(f"v_mul_f32 v{M}, %[b0], v{w_prev[0]}")
    if "c" in parts:
        out.append(f"v_sub_f32 v{T}, v{X[i % 2]}, v{a_cur[0]}")
    if "d" in parts:
        out.append(f"v_pk_mul_f32 {pair(BP[i % 4])}, %[ab2], {pair(w_prev)} op_sel_hi:[1,0]")
    if "e" in parts:
        out.append(f"v_sub_f32 v{w_cur[0]}, v{T}, v{BP[(i + 3) % 4][0]}")
    if "f" in parts:
        out.append(f"v_add_f32 v{YV[(i - 2) % 16]}, v{S[i % 2]}, v{BP[(i + 1) % 4][1]}")
    if "g" in parts:
        out.append(f"v_add_f32 v{S[(i + 1) % 2]}, v{M}, v{a_prev[1]}")
    if "h" in parts:
        if first_use_wait:
            out.append("s_waitcnt lgkmcnt(6)")
        out.append(f"v_cndmask_b32_dpp v{X[(i + 1) % 2]}, v{YV[(i - 3) % 16]}, v{XV[(i + 1) % 16]}, vcc "
                   "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")


def quad(regs, q):
    return f"v[{regs[4 * q]}:{regs[4 * q + 3]}]"


def batch_steps(out, steps, prefetch):
    """steps of one batch in walk order with the LDS traffic that rides along"""
    for i in steps:
        ii = i % 16
        step(i, out, first_use_wait=prefetch and ii in (15, 3, 7, 11))
        if ii in (2, 6, 10, 14) and prefetch:       # the quad of samples that is dead now <- the next batch's
            q = (ii - 2) // 4
            out.append(f"ds_read_b128 {quad(XV, q)}, v{AIN} offset:{64 + 16 * q}")
            if q == 3:
                out.append(f"v_add_u32 v{AIN}, 64, v{AIN}")
        if ii in (5, 9, 13):                        # y[4q .. 4q+3] complete
            q = (ii - 5) // 4
            out.append(f"ds_write_b128 v{AOUT}, {quad(YV, q)} offset:{16 * q}")
        if ii == 1 and i >= 16:                     # y[12..15] of the batch that just ended
            out.append(f"ds_write_b128 v{AOUT}, {quad(YV, 3)} offset:48")
            out.append(f"v_add_u32 v{AOUT}, v{AINC}, v{AOUT}")


def generate():
    o = []
    # ---- entry
    o.append("s_mov_b64 vcc, %[q0]")
    o.append(f"v_mov_b32 v{AIN}, %[ain]")
    o.append(f"v_mov_b32 v{AOUT}, %[aout]")
    o.append(f"v_mov_b32 v{AINC}, %[ainc]")
    for q in range(4):
        o.append(f"ds_read_b128 {quad(XV, q)}, v{AIN} offset:{16 * q}")
    o.append(f"v_mov_b32 v{WP[0][0]}, %[w2]")
    o.append(f"v_mov_b32 v{WP[1][0]}, %[w1]")
    o.append(f"v_mov_b32 v{TP[0]}, %[w3]")
    o.append(f"v_mov_b32 v{YV[12]}, %[y3]")                     # y[-4]: the left lane's value for x[0]
    o.append(f"v_mov_b32 v{YV[13]}, %[y2]")                     # y[-3]: for x[1]
    o.append(f"v_mov_b32 v{S[0]}, %[y1]")                       # "s[-2]" = y[-2] ...
    o.append(f"v_mov_b32 v{BP[1][1]}, 0x80000000")              # ... + (-0.0)
    o.append(f"v_pk_mul_f32 {pair(AP[1])}, %[ab1], {pair(WP[0])} op_sel_hi:[1,0]")   # A[-1] = {a1, b1} w[-2]
    o.append(f"v_pk_mul_f32 {pair(BP[3])}, %[ab2], {pair(WP[0])} op_sel_hi:[1,0]")   # B[-1] = {a2, b2} w[-2]
    o.append(f"v_pk_mul_f32 {pair(BP[2])}, %[ab2], {pair(TP)} op_sel_hi:[1,0]")      # B[-2] = {a2, b2} w[-3]
    o.append("s_waitcnt lgkmcnt(0)")
    o.append(f"v_cndmask_b32_dpp v{X[0]}, v{YV[12]}, v{XV[0]}, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
    batch_steps(o, [0, 1], prefetch=False)
    o.append("s_cmp_eq_u32 %[nb], 1")
    o.append("s_cbranch_scc1 2f")
    # ---- nb - 1 rotated batches: steps 2..15 of this batch, 0', 1' of the next
    o.append("1:")
    batch_steps(o, list(range(2, 18)), prefetch=True)
    o.append("s_add_i32 %[nb], %[nb], -1")
    o.append("s_cmp_lg_u32 %[nb], 1")
    o.append("s_cbranch_scc1 1b")
    # ---- the last batch: steps 2..15, then the output half of steps 14 and 15
    o.append("2:")
    tail = []
    batch_steps(tail, list(range(2, 14)), prefetch=False)
    o += tail
    o.append(f"v_mov_b32 %[w3], v{WP[1][0]}")                   # w[13]
    batch_steps(o, [14, 15], prefetch=False)
    step(16, o, parts="bfg")
    step(17, o, parts="f")
    o.append(f"ds_write_b128 v{AOUT}, {quad(YV, 3)} offset:48")
    o.append(f"v_mov_b32 %[w1], v{WP[1][0]}")                   # w[15]
    o.append(f"v_mov_b32 %[w2], v{WP[0][0]}")                   # w[14]
    for k in range(4):
        o.append(f"v_mov_b32 %[y{k}], v{YV[15 - k]}")
    return o


def main():
    lines = generate()
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "graphaudio_amd", "csrc", "ga_biquad_pipe_asm.inc")
    with open(path, "w") as f:
        f.write("// generated by tools/gen_biquad_pipe_asm.py -- do not edit; see there for the schedule\n")
        for ln in lines:
            f.write(f'"{ln}\\n"\n')
    with open(path.replace("_asm.inc", "_asm_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_biquad_pipe_asm.py -- do not edit\n")
        f.write(", ".join(f'"v{r}"' for r in CLOBBER) + ', "vcc", "scc", "memory"\n')
    print(f"{len(lines)} lines -> {os.path.normpath(path)}")


if __name__ == "__main__":
    main()
