"""Throughput of the media parity scenes at the benchmark resolution (1280x720), uniform sampler, adaptive off.
Not a bench.py line (BASELINE.json's metric is quoted on the Cornell box without media); recorded in profiles/."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import tungsten_amd as tg  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tmp = tempfile.mkdtemp(prefix="tg_media_")
for name, edit in (("cornell", None), ("cornell_fog", scenes._fog), ("cornell_smoke", scenes._smoke), ("cornell_fog_smoke", scenes._fog_and_smoke),
                   ("cornell_expfog", scenes._expfog), ("cornell_expfog_smoke", scenes._expfog_and_smoke)) + (
                  (("cornell_atmosphere", scenes._atmosphere), ("cornell_atmosphere_smoke", scenes._atmosphere_and_smoke)) if os.environ.get("TG_MEDIA_ATMOSPHERE", "1") != "0" else ()):
    warm = tg.Renderer(scenes.cornell(tmp, name=name + "_warm.json", resolution=(1280, 720), spp=4, edit=edit), seed=tg.DEFAULT_SEED)
    warm.render()                    # warm-up: device context, code objects, allocations
    warm.close()
    path = scenes.cornell(tmp, name=name + ".json", resolution=(1280, 720), spp=spp, edit=edit)
    r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
    for kv in sys.argv[2:]:          # option=value pairs, e.g. media_lean=0
        r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    secs = r.render()
    c = r.counters()
    r.close()
    print(json.dumps({"scene": name, "spp": spp, "seconds": secs, "msamples_per_s": 1280*720*spp/secs/1e6,
                      "closest_rays_per_sample": c.closest_rays/max(c.samples, 1), "shadow_rays_per_sample": c.shadow_rays/max(c.samples, 1)}))
