"""Static invariants of the built gfx950 code objects (no GPU needed): the properties DESIGN.md section 6 found to matter
for the NeRF-MLP kernel -- no scratch in the kernel (a scratch reload waits on vmcnt, i.e. on the weight DMA in flight)
and no compiler-inserted full DMA wait per slab step -- are checked on the ISA, so a toolchain or source change that
brings either back fails here and not as a silent 3-10 % loss on the GPU box."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nerf_sampling_amd", "libnerf_sampling_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _gfx950_code_objects(path):
    """Every gfx950 device ELF in the library's clang offload bundles (one bundle per translation unit)."""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        base = blob.find(MAGIC, pos)
        if base < 0:
            return out
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[base + off:base + off + size])
        pos = base + len(MAGIC)


def _isa_of(kernel_name: bytes):
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    for co in _gfx950_code_objects(LIB):
        if kernel_name not in co or (kernel_name == b"nerf_mlp_ob16_kernel" and b"ELb1ELi5EEE" in co):
            continue                     # (the five-tile unit of ns_nerf_mlp_ob16.hip has its own test)
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name],
                                 capture_output=True, text=True, check=True).stdout
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name],
                                   capture_output=True, text=True, check=True).stdout
        return dis, notes
    pytest.fail(f"no gfx950 code object with {kernel_name.decode()} in the library")


@pytest.fixture(scope="module")
def nerf16_isa():
    return _isa_of(b"nerf_mlp_ob16_kernel")


def _functions(dis):
    """name -> instruction lines"""
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
        elif cur is not None and line.startswith("\t"):
            cur.append(line.strip())
    return out


def test_nerf16_kernels_use_no_scratch_and_no_full_dma_wait(nerf16_isa):
    dis, notes = nerf16_isa
    fns = {k: v for k, v in _functions(dis).items() if "nerf_mlp_ob16_kernel" in k}
    # the rays -> raw kernels (EMBEDDED = false: mangled "...ELb0ELb<PROD>ELi<tiles>EEE"), bf16 and f16: the generic program at
    # W = 256 and 128, and the production program (hand-scheduled hidden layers) at W = 256
    main = {k: v for k, v in fns.items() if re.search(r"ELi[48]ELb0ELb[01]ELi4EEE", k)}
    assert len(main) == 6, sorted(fns)
    assert len(fns) == 12, sorted(fns)
    for name, ins in main.items():
        text = "\n".join(ins)
        assert "scratch_" not in text, f"{name}: scratch access in the kernel"
        mfma = sum("v_mfma_f32_16x16x32" in i for i in ins)
        full_waits = sum(bool(re.search(r"s_waitcnt vmcnt\(0\)(?! *lgkmcnt)|s_waitcnt vmcnt\(0\)$", i)) for i in ins)
        dma = sum("global_load_lds_dwordx4" in i for i in ins)
        assert mfma > 1000 and dma > 50
        # the only full VMEM waits left are outside the slab loop (first-group staging, final drain); with the builtin
        # LDS-DMA the compiler emitted one per slab step (92 in this kernel)
        assert full_waits <= 10, f"{name}: {full_waits} s_waitcnt vmcnt(0)"
    # kernel descriptors agree: no private segment for those kernels
    for name in main:
        m = re.search(re.escape(name) + r".*?\.private_segment_fixed_size:\s*(\d+)", notes, re.S)
        if m:
            assert int(m.group(1)) == 0


def test_production_kernel_is_straight_line_hand_scheduled_code(nerf16_isa):
    """The production program (8 x 256, skips = [4]): every MFMA of a group pass appears exactly once (no loop over layers:
    4 180 per 64 samples), and between the ten generated layer statements the compiler moves no
    activation register (the statements pin the two activation sets to a[0:127] / v[128:255]; a copy there would be 128
    v_accvgpr / v_mov instructions per layer)."""
    dis, _ = nerf16_isa
    fns = {k: v for k, v in _functions(dis).items() if re.search(r"Mma16BF16ELi8ELb0ELb1ELi4EEE", k)}
    assert len(fns) == 1, sorted(fns)
    ins = [i.split("//")[0].strip() for i in next(iter(fns.values()))]
    assert sum("v_mfma_f32_16x16x32_bf16" in i for i in ins) == 4180
    # a generated statement opens with its first bias read into v[48:51]
    # ten statements: layer 0, seven hidden layers, the view layer, the rgb head
    starts = [n for n, i in enumerate(ins) if re.match(r"ds_read_b128 v\[48:51\], v\d+$", i)]
    assert len(starts) == 10, starts
    mfmas = []
    for a, b in zip(starts, starts[1:] + [len(ins)]):
        seg = ins[a:b]
        if b == len(ins):                # the last statement ends where it restores M0
            seg = seg[:next(n for n, i in enumerate(seg) if i.startswith("s_mov_b32 m0,")) + 1]
        mfmas.append(sum("v_mfma" in i for i in seg))
        copies = sum(i.startswith(("v_accvgpr_mov", "v_mov_b32", "v_mov_b64", "v_accvgpr_read")) for i in seg)
        assert copies <= 16, (a, b, copies)      # (a full activation set would be 128)
        # what the statements themselves need: one v_accvgpr_write per dword written to set A, none in the A -> V layers
        w = sum(i.startswith("v_accvgpr_write") for i in seg)
        assert min(abs(w - k) for k in (0, 64, 128)) <= 8, (a, b, w)
        # hazard pads are the exception in the hand-written stream (the compiled layers carry ~0.2 s_nop per MFMA)
        assert sum(i.startswith("s_nop") for i in seg) <= 12
    assert mfmas == [128, 512, 512, 512, 512, 640, 512, 512, 324, 16], mfmas


def test_five_tile_production_kernel_keeps_scratch_out_of_the_layers():
    """The five-tile production kernels (the file's second translation unit, NS_OB16_TU_T5): 5 225 MFMAs per 80-sample wave
    pass, no activation copies between the seven generated statements -- and the few registers the compiler spills around
    them (the statements leave it v[64:95]) are stored / reloaded only before the first and after the last statement, i.e.
    never while the weight ring is being walked by hand-counted vmcnt waits."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    found = None
    for co in _gfx950_code_objects(LIB):
        if b"nerf_mlp_ob16_kernel" in co and b"ELb1ELi5EEE" in co:
            found = co
    assert found is not None, "no code object with the five-tile kernels"
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(found)
        f.flush()
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    fns = {k: v for k, v in _functions(dis).items() if "nerf_mlp_ob16_kernel" in k}
    assert len(fns) == 4 and all("ELb1ELi5EEE" in k for k in fns), sorted(fns)     # bf16 / f16 x rays / pre-embedded
    for name, raw in fns.items():
        ins = [i.split("//")[0].strip() for i in raw]
        assert sum("v_mfma_f32_16x16x32" in i for i in ins) == 5225, name
        starts = [n for n, i in enumerate(ins) if re.match(r"ds_read_b128 v\[56:59\], v\d+$", i)]
        assert len(starts) == 10, (name, starts)
        last_end = next(n for n, i in enumerate(ins[starts[-1]:], starts[-1]) if i.startswith("s_mov_b32 m0,"))   # a statement restores M0 last
        for n, i in enumerate(ins):
            if i.startswith("scratch_"):
                assert n < starts[0] or n > last_end, (name, n, i)
        for a, b in zip(starts, starts[1:]):
            seg = ins[a:b]
            assert sum(i.startswith(("v_accvgpr_mov", "v_mov_b32", "v_mov_b64", "v_accvgpr_read")) for i in seg) <= 24   # (a full set is 160)
            w = sum(i.startswith("v_accvgpr_write") for i in seg)      # a statement's own AGPR writes (+ a few strays: the
            assert min(abs(w - k) for k in (0, 80, 160)) <= 12, w         # compiler parks single values in the free a[200:255])


def _check_mlp_kernels(dis, notes, name_part, n_expected, mfma_pat, min_mfma):
    fns = {k: v for k, v in _functions(dis).items() if name_part in k}
    assert len(fns) == n_expected, sorted(fns)
    for name, ins in fns.items():
        text = "\n".join(ins)
        assert "scratch_" not in text, f"{name}: scratch access in the kernel"
        assert sum(mfma_pat in i for i in ins) > min_mfma, name
        full_waits = sum(bool(re.search(r"s_waitcnt vmcnt\(0\)(?! *lgkmcnt)|s_waitcnt vmcnt\(0\)$", i)) for i in ins)
        assert full_waits <= 10, f"{name}: {full_waits} s_waitcnt vmcnt(0)"
        m = re.search(re.escape(name) + r".*?\.private_segment_fixed_size:\s*(\d+)", notes, re.S)
        if m:
            assert int(m.group(1)) == 0


def test_depthnet16_kernels_use_no_scratch_and_no_full_dma_wait():
    """The folded DepthNet on the same engine: bf16, f16 and split-f16 (f16x3), W = 256 and 128 -- no scratch (round 1's
    kernel spilled 91 VGPRs and kept a 655 MB global stash), no compiler-inserted full DMA wait in the slab loop."""
    dis, notes = _isa_of(b"depthnet_ob16_kernel")
    _check_mlp_kernels(dis, notes, "depthnet_ob16_kernel", 8, "v_mfma_f32_16x16x32", 500)


def test_depthnet_production_kernel_is_straight_line_generated_code():
    """The production DepthNet (ten 8-K-block -> 256 LeakyReLU layers after the fold, fp16 operands): every layer is one
    generated statement (tools/gen_ob16_asm.py, act = "leaky"), 512 MFMAs each, the 1-row head is the only compiled layer;
    between the statements the compiler moves no activation set (a copy would be 128 v_accvgpr / v_mov per layer) and the
    kernel uses no scratch."""
    dis, notes = _isa_of(b"depthnet_ob16_kernel")
    fns = {k: v for k, v in _functions(dis).items() if re.search(r"Mma16F16EEELi8ELb1EEE", k)}
    assert len(fns) == 1, sorted(_functions(dis))
    ins = [i.split("//")[0].strip() for i in next(iter(fns.values()))]
    assert not any(i.startswith("scratch_") for i in ins)
    assert sum("v_mfma_f32_16x16x32_f16" in i for i in ins) == 10 * 512 + 32
    starts = [n for n, i in enumerate(ins) if re.match(r"ds_read_b128 v\[48:51\], v\d+$", i)]
    assert len(starts) == 10, starts
    for a, b in zip(starts, starts[1:]):
        seg = ins[a:b]
        assert sum("v_mfma" in i for i in seg) == 512
        assert sum(i.startswith("v_pk_mul_f16") for i in seg) == 128 and sum(i.startswith("v_pk_max_f16") for i in seg) == 128
        assert sum(i.startswith(("v_accvgpr_mov", "v_mov_b32", "v_mov_b64", "v_accvgpr_read")) for i in seg) <= 12
        w = sum(i.startswith("v_accvgpr_write") for i in seg)      # V -> A layers write set A one dword at a time
        assert min(abs(w - k) for k in (0, 128)) <= 8, w
        assert sum(i.startswith("s_nop") for i in seg) <= 12


def test_nerf_x3_kernels_use_no_scratch_and_no_full_dma_wait():
    dis, notes = _isa_of(b"nerf_mlp_x3_kernel")
    # rays -> raw kernels (EMBEDDED = false): the generic program at W = 256 / 128 and the production program (generated
    # hidden layers) at W = 256
    fns = {k: v for k, v in _functions(dis).items() if re.search(r"nerf_mlp_x3_kernelILi[48]ELb0ELb[01]EEE", k)}
    assert len(fns) == 3, sorted(fns)
    _check_mlp_kernels("\n".join(f"0000 <{k}>:\n" + "\n".join("\t" + i for i in v) for k, v in fns.items()), notes,
                       "nerf_mlp_x3_kernel", 3, "v_mfma_f32_16x16x32_f16", 1500)
    prod = [i.split("//")[0].strip() for k, v in fns.items() if "ILi8ELb0ELb1EEE" in k for i in v]
    assert sum("v_mfma_f32_16x16x32_f16" in i for i in prod) == 3 * 4180 // 2    # straight-line: three MFMAs per product term, two tiles
