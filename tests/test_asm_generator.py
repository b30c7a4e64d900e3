"""tools/gen_ob16_asm.py (no GPU needed): the committed csrc/ns_ob16_asm.inc is what the generator emits, and the
generator's independent checker pass really rejects streams that break a hazard rule or read an LDS result early."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "gen_ob16_asm.py")
INC = os.path.join(ROOT, "nerf_sampling_amd", "csrc", "ns_ob16_asm.inc")


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_ob16_asm", GEN)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_streams_are_the_generator_output(tmp_path):
    out = tmp_path / "x.inc"
    subprocess.run([sys.executable, GEN, "-o", str(out)], check=True, capture_output=True)
    assert out.read_text() == open(INC).read(), "regenerate: python tools/gen_ob16_asm.py"


@pytest.mark.parametrize("in_a", [True, False])
@pytest.mark.parametrize("skip", [False, True])
def test_stream_shape(gen, in_a, skip):
    e, slabs = gen.gen_layer("bf16", in_a, skip)
    nkb = 10 if skip else 8
    assert slabs == nkb
    kinds = [i.kind for i in e.ins]
    assert kinds.count("mfma") == 4 * 16 * nkb                      # four tiles x 16 sub-blocks x K-blocks
    assert kinds.count("dma") == 4 * slabs                          # every slab refilled by four pieces per wave
    assert sum("s_barrier" in i.text for i in e.ins) == slabs
    assert kinds.count("lds") == 16 * nkb + 16                      # one fragment per chunk + one bias tuple per sub-block
    valu = [i.text.split()[0] for i in e.ins if i.kind == "valu"]
    assert valu.count("v_cvt_pk_bf16_f32") == 128 and valu.count("v_pk_max_i16") == 128   # one per output dword
    assert valu.count("v_accvgpr_write_b32") == (0 if in_a else 128)
    # every accumulator chain is K-ordered: the MFMAs writing one accumulator read the K-blocks in ascending order
    gen.check(e.ins)


@pytest.mark.parametrize("in_a", [True, False])
@pytest.mark.parametrize("skip", [False, True])
def test_five_tile_stream_shape(gen, in_a, skip):
    """the five-tile wave (80 samples): 160 + 160 activation registers, ten dwords per sub-block through the same pipeline
    (two of them through gap 3 when a sub-block has only eight steps)"""
    e, slabs = gen.gen_layer("bf16", in_a, skip, gen.Map5)
    nkb = 10 if skip else 8
    kinds = [i.kind for i in e.ins]
    assert slabs == nkb and kinds.count("mfma") == 5 * 16 * nkb and kinds.count("dma") == 4 * slabs
    valu = [i.text.split()[0] for i in e.ins if i.kind == "valu"]
    assert valu.count("v_cvt_pk_bf16_f32") == 160 and valu.count("v_pk_max_i16") == 160
    assert valu.count("v_accvgpr_write_b32") == (0 if in_a else 160)
    written = set().union(*[i.writes for i in e.ins if i.kind == "valu" and i.text.startswith(("v_pk_max", "v_accvgpr_write"))])
    out = {("v", r) for r in range(96, 256)} if in_a else {("a", r) for r in range(160)}
    assert out <= written
    gen.check(e.ins)


@pytest.mark.parametrize("tiles", [4, 5])
def test_special_layer_streams(gen, tiles):
    """layer 0, the view layer (sigma sub-block returned raw) and the rgb head: streams that are not slab multiples -- pad
    steps without MFMAs, a short last slab that still gets its four refill pieces and its barrier"""
    m = gen.Map4 if tiles == 4 else gen.Map5
    want = {"layer0": (16 * 2, 2, True), "views": (9 * 9, 6, True), "rgb": (1 * 4, 1, False)}
    for kind, (chunks, slabs_want, converts) in want.items():
        e, slabs = gen.gen_layer_special("bf16", kind, m)
        kinds = [i.kind for i in e.ins]
        assert slabs == slabs_want
        assert kinds.count("mfma") == tiles * chunks
        assert kinds.count("dma") == 4 * slabs and sum("s_barrier" in i.text for i in e.ins) == slabs
        n_cvt = sum(i.text.startswith("v_cvt_pk") for i in e.ins)
        n_conv_sb = {"layer0": 16, "views": 8, "rgb": 0}[kind]          # converted sub-blocks (the view layer's last one is raw)
        assert n_cvt == 2 * tiles * n_conv_sb and (n_cvt > 0) == converts
        # every conversion lands in set A (an AGPR write per dword), K-blocks 0..7 (layer 0) or 0..3 (view layer)
        wr = set().union(*[i.writes for i in e.ins if i.text.startswith("v_accvgpr_write")]) if converts else set()
        kbs = {(r - 32 * (r // 32)) // 4 for f, r in wr}
        assert all(f == "a" for f, _ in wr) and kbs == ({0, 1, 2, 3, 4, 5, 6, 7} if kind == "layer0" else ({0, 1, 2, 3} if converts else set()))
        gen.check(e.ins)


@pytest.mark.parametrize("in_a", [True, False])
@pytest.mark.parametrize("skip", [False, True])
def test_split_operand_stream_shape(gen, in_a, skip):
    """f16x3: per K-block a W_hi chunk (x_hi and x_lo of both tiles) and a W_lo chunk (x_hi only): three MFMAs per product
    term; seven (nine with the AGPR writes) VALU instructions per converted dword pair: NaN-keeping ReLU by compare + select,
    one v_cvt_pk for the two fp16 hi halves, one mixed-precision fma per fp16 remainder"""
    e, slabs = gen.gen_layer_x3(in_a, skip)
    nkb = 10 if skip else 8
    kinds = [i.kind for i in e.ins]
    assert slabs == 2 * nkb and kinds.count("mfma") == 16 * nkb * 6 and kinds.count("lds") == 16 * 2 * nkb + 16
    valu = [i.text.split()[0] for i in e.ins if i.kind == "valu"]
    pairs = 16 * 4                                    # sub-blocks x (two tiles x two dword pairs)
    assert valu.count("v_cvt_pk_f16_f32") == pairs and valu.count("v_cmp_ngt_f32_e32") == 2 * pairs
    assert valu.count("v_fma_mixlo_f16") == pairs and valu.count("v_fma_mixhi_f16") == pairs and valu.count("v_sub_f32_e32") == 0
    assert valu.count("v_accvgpr_write_b32") == (0 if in_a else 2 * pairs)
    assert len(valu) - valu.count("v_lshl_add_u32") == pairs * (7 if in_a else 9)
    gen.check(e.ins)
    if not skip:      # the DepthNet's variant: LeakyReLU on the fp32 values instead of the ReLU, same count
        e2, slabs2 = gen.gen_layer_x3(in_a, False, act="leaky")
        valu2 = [i.text.split()[0] for i in e2.ins if i.kind == "valu"]
        assert slabs2 == slabs and valu2.count("v_mul_f32_e32") == 2 * pairs and valu2.count("v_max_f32_e32") == 2 * pairs
        assert valu2.count("v_cmp_ngt_f32_e32") == 0 and len(valu2) == len(valu)
        gen.check(e2.ins)


@pytest.mark.parametrize("in_a", [True, False])
def test_leaky_stream_shape(gen, in_a):
    """the DepthNet's fp16 hidden layers: LeakyReLU(0.01) on the packed value, one more pipeline stage (v_pk_mul_f16 by the
    slope operand, v_pk_max_f16) than the ReLU streams, same MFMAs and ring protocol"""
    e, slabs = gen.gen_layer("f16", in_a, False, act="leaky")
    kinds = [i.kind for i in e.ins]
    assert slabs == 8 and kinds.count("mfma") == 4 * 16 * 8 and kinds.count("dma") == 4 * slabs
    valu = [i.text.split()[0] for i in e.ins if i.kind == "valu"]
    assert valu.count("v_cvt_pk_f16_f32") == 128 and valu.count("v_pk_mul_f16") == 128 and valu.count("v_pk_max_f16") == 128
    assert valu.count("v_pk_max_i16") == 0 and valu.count("v_accvgpr_write_b32") == (0 if in_a else 128)
    assert all("%[slope]" in i.text for i in e.ins if i.text.startswith("v_pk_mul_f16"))
    gen.check(e.ins)
    with pytest.raises(AssertionError):
        gen.gen_layer("bf16", in_a, False, act="leaky")          # no packed bf16 multiply on gfx950


def test_checker_rejects_broken_streams(gen):
    e, _ = gen.gen_layer("bf16", True, False)
    ins = list(e.ins)
    # (1) a conversion moved right behind the MFMA that produces its input
    k = next(n for n, i in enumerate(ins) if i.kind == "valu" and i.text.startswith("v_cvt_pk"))
    src = ins[k].reads
    w = max(n for n in range(k) if ins[n].kind == "mfma" and ins[n].writes & src)
    bad = ins[:w + 1] + [ins[k]] + ins[w + 1:k] + ins[k + 1:]
    with pytest.raises(AssertionError, match="MFMA D->valu|clock model"):
        gen.check(bad)
    # (2) a fragment wait dropped
    k = next(n for n, i in enumerate(ins) if i.kind == "wait" and n > 40)
    with pytest.raises(AssertionError, match="LDS result not waited for"):
        gen.check(ins[:k] + ins[k + 1:])
    # (3) a VALU write directly in front of the MFMA that reads it
    k = next(n for n, i in enumerate(ins) if i.kind == "valu" and i.text.startswith("v_pk_max"))
    dst = ins[k].writes
    fake = gen.Ins("v_mfma_f32_16x16x32_bf16 v[0:3], v[32:35], v[128:131], v[0:3]", "mfma",
                   reads=(gen.R('v', 32, 4), next(gen.R(f, b, 1) for f, b in dst)), writes=(gen.R('v', 0, 4),), creads=(gen.R('v', 0, 4),))
    with pytest.raises(AssertionError, match="VALU->MFMA"):
        gen.check(ins[:k + 1] + [fake])
