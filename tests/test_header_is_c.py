"""include/te_msm.h is a C header: the reference-side bindings (cgo-style FFI generators, N-API addons written in C) include it
from C.  Compiles a C translation unit that uses every declared entry point's type, with gcc -std=c99 -pedantic."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_compiles_as_c99(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    hdr = open(os.path.join(ROOT, "include", "te_msm.h")).read()
    names = sorted(set(re.findall(r"\b(te_msm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30 and "te_msm_submit_async" in names and "te_msm_ticket_device" in names
    src = tmp_path / "use.c"
    src.write_text('#include "te_msm.h"\n#include <stddef.h>\n' +
                   "void* table[] = {%s};\n" % ", ".join("(void*)(size_t)&%s" % n for n in names) +
                   "int main(void) { return sizeof table > 0 ? 0 : 1; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-c", "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "use.o"), str(src)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
