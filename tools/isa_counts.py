"""Static instruction mix of one kernel of a built library:  python tools/isa_counts.py LIB KERNEL_SUBSTRING"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernel_invariants as t  # noqa: E402

lib, pat = sys.argv[1], sys.argv[2]
for co in t._gfx950_code_objects(lib):
    if pat.split("I")[0].encode() not in co and pat.encode() not in co:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        dis = subprocess.run([os.path.join(t.LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout
    for k, ins in t._functions(dis).items():
        if pat not in k:
            continue
        n = lambda p: sum(bool(re.search(p, i)) for i in ins)  # noqa: E731
        print(f"{os.path.basename(lib)} {k[:60]}: mfma {n('v_mfma')} valu {n(r'^v_(?!mfma)')} (v_mov {n('^v_mov')} accvgpr {n('v_accvgpr')} "
              f"cvt_pk {n('v_cvt_pk')} pk_max {n('v_pk_max')}) salu {n(r'^s_')} (waitcnt {n('s_waitcnt')} nop {n('s_nop')} barrier {n('s_barrier')}) "
              f"ds_read {n('ds_read')} ds_write {n('ds_write')} dma {n('global_load_lds')} scratch {n('scratch_')}")
