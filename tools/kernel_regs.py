"""Register / scratch / spill figures of every kernel in the built library (reads the gfx950 code-object notes)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernel_invariants import _gfx950_code_objects, LIB, LLVM  # noqa: E402

pat = sys.argv[1] if len(sys.argv) > 1 else ""
for co in _gfx950_code_objects(LIB):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if pat in name:
            print(f"{name[:90]:90s} agpr {blk.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} "
                  f"scratch {g('private_segment_fixed_size'):>4s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size')}")
