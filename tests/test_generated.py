"""Generated sources are committed (the build does not run generators): they have to match what their generators write."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_biquad_pipe_assembly_matches_its_generator():
    spec = importlib.util.spec_from_file_location("gen_biquad_pipe_asm", os.path.join(ROOT, "tools", "gen_biquad_pipe_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    lines = gen.generate()
    body = open(os.path.join(ROOT, "graphaudio_amd", "csrc", "ga_biquad_pipe_asm.inc")).read().splitlines()
    quoted = [ln for ln in body if ln.startswith('"')]
    assert quoted == ['"%s\\n"' % ln for ln in lines]
    # 16 steps x 8 vector instructions in the loop body, none of them right behind its producer (the schedule's whole point)
    loop = lines[lines.index("1:") + 1:lines.index("s_add_i32 %[nb], %[nb], -1")]
    valu = [ln for ln in loop if ln.startswith("v_") and not ln.startswith("v_add_u32")]
    assert len(valu) == 16 * 8
    clob = open(os.path.join(ROOT, "graphaudio_amd", "csrc", "ga_biquad_pipe_asm_clobbers.inc")).read()
    for r in gen.CLOBBER:
        assert '"v%d"' % r in clob

    def dst(ln):
        return ln.split()[1].rstrip(",")

    def regs(tok):   # "v[228:229]" -> {228, 229} ; "v240" -> {240}
        tok = tok.strip(",")
        if tok.startswith("v["):
            a, b = tok[2:-1].split(":")
            return set(range(int(a), int(b) + 1))
        if tok.startswith("v") and tok[1:].isdigit():
            return {int(tok[1:])}
        return set()
    for prev, cur in zip(valu, valu[1:]):
        written = regs(dst(prev))
        read = set()
        for tok in cur.split()[2:]:
            read |= regs(tok)
        assert not (written & read), (prev, cur)
