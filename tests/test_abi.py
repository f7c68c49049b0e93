"""The C-ABI boundary (include/tungsten_hip.h, include/tungsten_host.h): the library loads, exports every
declared symbol, the ctypes mirror has the C layout, and -- on a box without a GPU -- every device
entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

import tungsten_amd as tg
from tungsten_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("tungsten_hip.h", "tungsten_host.h")]


def declared_functions():
    names = []
    for h in HEADERS:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        inline = set(re.findall(r"static\s+inline\s+\w+\s+(tghip_[a-z_]+)\s*\(", src))   # header-only helpers (tghip_tile_owner): no symbol
        names += [n for n in re.findall(r"\b(tghip_[a-z_]+|tgh_[a-z_]+)\s*\(", src) if n not in inline]
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 25
    lib = C.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libtungsten_hip.so does not export %s" % n
        assert n in capi.PROTOTYPES, "tungsten_amd/capi.py has no prototype for %s" % n
    assert sorted(capi.PROTOTYPES) == names


def test_product_library_does_not_link_the_oracle():
    out = subprocess.check_output(["ldd", capi.LIB_PATH]).decode()
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    assert "oracle_" not in syms


def test_ctypes_layout_matches_the_headers(tmp_path):
    structs = ["TgHipBvhNode", "TgHipWideNode", "TgHipPrimRec", "TgHipTriAttr", "TgHipObject", "TgHipBsdf", "TgHipTexture", "TgHipMedium", "TgHipCamera",
               "TgHipSettings", "TgHipSceneDesc", "TgHipPassDesc", "TgHipAuxPixel", "TgHipCounters", "TgHipRay", "TgHipHit", "TgHostSceneInfo"]
    src = '#include <stdio.h>\n#include "tungsten_host.h"\nint main(void){\n'
    for s in structs:
        src += 'printf("%s %%zu\\n", sizeof(%s));\n' % (s, s)
    src += 'printf("camera %zu\\n", offsetof(TgHipSceneDesc, camera));\nreturn 0;}\n'
    c = tmp_path/"sz.c"
    c.write_text(src)
    exe = str(tmp_path/"sz")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", exe])
    got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for s in structs:
        assert int(got[s]) == C.sizeof(getattr(capi, s)), s
    assert int(got["camera"]) == capi.TgHipSceneDesc.camera.offset
    assert C.sizeof(capi.TgHipWideNode) == 80
    assert C.sizeof(capi.TgHipBvhNode) == 64 and C.sizeof(capi.TgHipPrimRec) == 48 and C.sizeof(capi.TgHipTriAttr) == 64
    assert C.sizeof(capi.TgHipRay) == 32 and C.sizeof(capi.TgHipHit) == 16


def test_error_paths_without_arguments():
    lib = tg.lib
    assert lib.tghip_upload_scene(None, None) == -1
    assert lib.tghip_render_pass(None, None) == -1
    assert lib.tghip_wait(None) == -1
    assert lib.tghip_set_option(None, b"x", 1) == -1
    lib.tghip_destroy(None)   # no-op
    with pytest.raises(tg.TungstenError):
        tg.FlattenedScene("/nonexistent/scene.json")


@pytest.mark.skipif(tg.device_count() > 0, reason="this is the no-GPU contract")
def test_no_device_means_loud_failure_not_a_cpu_fallback(tmp_path):
    import scenes
    assert tg.lib.tghip_create(0) is None
    assert b"no HIP device" in tg.lib.tghip_last_error(None)
    p = scenes.cornell(tmp_path, resolution=(32, 18), spp=1)
    with pytest.raises(tg.TungstenError) as e:
        tg.Renderer(p)
    assert "no HIP device" in str(e.value)
