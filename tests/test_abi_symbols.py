"""CPU checks of the C-ABI boundary: libdartk.so loads, exports every function include/dartk.h declares, and the ctypes
mirrors of the public structs have the C layout. No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from dart_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _abi.lib()
    names = _abi.declared_symbols()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and every symbol has a ctypes signature (argtypes set) so calls are type-checked
    unsigned = [n for n in names if getattr(L, n).argtypes is None and n not in ("dartk_last_error", "dartk_version")]
    assert not unsigned, unsigned


def test_struct_layouts_match_the_header():
    """sizeof of the public structs as the C compiler sees them == the ctypes mirrors."""
    src = r'''
#include <stdio.h>
#include "dartk.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu\n", sizeof(dartk_da_point), sizeof(dartk_model_desc), sizeof(dartk_opt_options),
           sizeof(dartk_contact_prior), sizeof(dartk_noobs_options));
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(_abi.DaPoint), C.sizeof(_abi.ModelDesc), C.sizeof(_abi.OptOptions), C.sizeof(_abi.ContactPriorC),
                     C.sizeof(_abi.NoObsOptions)]


def test_header_cites_the_reference_for_every_entry_point():
    txt = open(os.path.join(ROOT, "include", "dartk.h")).read()
    assert len(re.findall(r"\.(cu|cpp|h):\d+", txt)) >= 40
