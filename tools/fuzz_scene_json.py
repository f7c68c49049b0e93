"""Robustness of the host loader (csrc/host/Scene.cpp, Json.hpp): the Cornell box's scene description with one to three random edits (junk values, deleted keys, appended
elements), each loaded in a child process -- it must load or raise, never crash.   python tools/fuzz_scene_json.py SEED COUNT   (CPU only)"""
import json, os, random, subprocess, sys, tempfile
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import scenes
tmp=tempfile.mkdtemp()
base=json.load(open(scenes.cornell(tmp, name='base.json', resolution=(32,18), spp=1)))
rng=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
JUNK=[None, True, -1, 0, 1e30, -1e30, float('nan'), "", "x", [], {}, [1], [1,2], [1,2,3,4], {"type":"nope"}, "lambert", 2**40, [[]], {"a":{}}]
def paths(x, p=()):
    yield p
    if isinstance(x, dict):
        for k,v in x.items(): yield from paths(v, p+(k,))
    elif isinstance(x, list):
        for i,v in enumerate(x): yield from paths(v, p+(i,))
def mutate(doc):
    doc=json.loads(json.dumps(doc))
    ps=[p for p in paths(doc) if p]
    for _ in range(rng.randint(1,3)):
        p=rng.choice(ps)
        cur=doc
        try:
            for k in p[:-1]: cur=cur[k]
            op=rng.random()
            if op<0.6: cur[p[-1]]=rng.choice(JUNK)
            elif op<0.8 and isinstance(cur, dict): del cur[p[-1]]
            elif isinstance(cur, list): cur.append(rng.choice(JUNK))
            else: cur[p[-1]]=rng.choice(JUNK)
        except Exception: pass
    return doc
CHILD='''
import sys
sys.path.insert(0,'/root/repo')
import tungsten_amd as tg
try:
    f=tg.FlattenedScene(sys.argv[1]); f.close(); print("ok")
except Exception as e:
    print("exc", type(e).__name__, str(e)[:80])
'''
bad=0
for i in range(int(sys.argv[2]) if len(sys.argv)>2 else 200):
    doc=mutate(base)
    path=os.path.join(tmp,'m%d.json'%i)
    try:
        s=json.dumps(doc)
    except Exception: continue
    s=s.replace('NaN','nan') if rng.random()<0.1 else s
    open(path,'w').write(s)
    r=subprocess.run([sys.executable,'-c',CHILD,path],stdout=subprocess.PIPE,stderr=subprocess.PIPE,universal_newlines=True,timeout=60)
    if r.returncode!=0:
        bad+=1
        print('CRASH rc',r.returncode,path, r.stderr[-300:])
        os.rename(path, '/tmp/crash_%d.json'%i)
print('done, crashes:',bad)
