"""Robustness of the host's decoders (csrc/host/ImageIO.cpp: own inflate / PNG / Radiance .hdr; Scene.cpp: .wo3 meshes): a scene whose texture, environment map or
mesh file carries random byte edits, truncations and insertions, each loaded in a child process -- it must load or raise, never crash.
   python tools/fuzz_scene_assets.py SEED COUNT   (CPU only; needs assets/materialtest)"""
import json, os, random, subprocess, sys, tempfile, shutil, struct, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import scenes
import numpy as np
rng=random.Random(int(sys.argv[1]))
N=int(sys.argv[2])
tmp=tempfile.mkdtemp()
# a small valid PNG (RGBA 8x8) written by hand
def png(w,h,ct=6):
    raw=b''.join(b'\x00'+bytes(rng.randrange(256) for _ in range(w*(4 if ct==6 else 3))) for _ in range(h))
    def chunk(t,d): return struct.pack('>I',len(d))+t+d+struct.pack('>I',zlib.crc32(t+d)&0xffffffff)
    return b'\x89PNG\r\n\x1a\n'+chunk(b'IHDR',struct.pack('>IIBBBBB',w,h,8,ct,0,0,0))+chunk(b'IDAT',zlib.compress(raw))+chunk(b'IEND',b'')
hdr=open('/root/repo/assets/materialtest/envmap.hdr','rb').read()[:20000]
wo3=open('/root/repo/assets/materialtest/Ring.wo3','rb').read()[:30000]
def mutate(b):
    b=bytearray(b)
    for _ in range(rng.randint(1,8)):
        op=rng.random()
        if op<0.5 and b: b[rng.randrange(len(b))]=rng.randrange(256)
        elif op<0.7 and len(b)>8: del b[rng.randrange(len(b)):][:rng.randint(1,64)]; 
        elif op<0.85: b=b[:rng.randrange(len(b)+1)]
        else: i=rng.randrange(len(b)+1); b[i:i]=bytes(rng.randrange(256) for _ in range(rng.randint(1,16)))
    return bytes(b)
CHILD='''
import sys
sys.path.insert(0,'/root/repo')
import tungsten_amd as tg
try:
    f=tg.FlattenedScene(sys.argv[1]); f.close(); print("ok")
except Exception as e:
    print("exc", type(e).__name__, str(e)[:80])
'''
def tex(scene):
    scene["bsdfs"][0]["albedo"]={"type":"bitmap","file":"t.png"}
    scene["primitives"].append({"name":"env","type":"infinite_sphere","emission":"e.hdr","transform":{}})
    scene["primitives"].append({"name":"m","type":"mesh","file":"m.wo3","bsdf":scene["bsdfs"][0]["name"],"transform":{}})
bad=0; oks=0
for i in range(N):
    d=os.path.join(tmp,'c%d'%i); os.makedirs(d)
    path=scenes.cornell(d, name='s.json', resolution=(16,9), spp=1, edit=tex)
    kind=rng.choice(['png','hdr','wo3'])
    open(os.path.join(d,'t.png'),'wb').write(mutate(png(8,8,rng.choice([2,6]))) if kind=='png' else png(8,8))
    open(os.path.join(d,'e.hdr'),'wb').write(mutate(hdr) if kind=='hdr' else open('/root/repo/assets/materialtest/envmap.hdr','rb').read())
    open(os.path.join(d,'m.wo3'),'wb').write(mutate(wo3) if kind=='wo3' else open('/root/repo/assets/materialtest/Ring.wo3','rb').read())
    r=subprocess.run([sys.executable,'-c',CHILD,path],stdout=subprocess.PIPE,stderr=subprocess.PIPE,universal_newlines=True,timeout=120)
    if r.returncode!=0:
        bad+=1; print('CRASH',kind,'rc',r.returncode,d,r.stderr[-200:])
        shutil.copytree(d,'/tmp/crash_asset_%s_%d'%(sys.argv[1],i))
    elif r.stdout.startswith('ok'): oks+=1
    shutil.rmtree(d,ignore_errors=True)
print('done, crashes:',bad,'loaded ok:',oks,'of',N)
