import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gen_golden as G
from pbrpathtracer_amd import scenes as S
out=sys.argv[1]; os.makedirs(out,exist_ok=True); count=int(sys.argv[2])
texts=list(G.OBJ_VARIANTS.values())
tokens=["f 1 2 99","f 0 1 2","f -9 1 2","f 1/9/1 2/1/1 3/1/1","f 1/1/9 2/1/1 3/1/1","f 1//-5 2//1 3//1","f 2147483647 1 2","f -2147483648 1 2","f 99999999999999 1 2","f 1 2 3 4 5 6 7 8 9 10 11 12",
        "f a b c","f 1/ 2/ 3/","f /1/1 2 3","v nan inf -inf","v 1e999 0 0","v","vt","vn 1","f","f 1","g","o","s -3","usemtl","\x00\x01\x02","f 1.5 2.5 3.5","f 1/2/3/4 2 3","f 5 5 5 5 5","f 1 1 2 2 3 3"]
pts, sc, _ = S.build_config("C3", out, width=64, height=36, nu=6, nv=4, tex_size=8)
ptsdata=open(pts,'rb').read()
for seed in range(count):
    rng=np.random.default_rng(seed)
    if seed%4==3:
        d=bytearray(ptsdata); m=seed%3
        if m==0: d=d[:int(rng.integers(0,len(d)))]
        elif m==1:
            for _ in range(int(rng.integers(1,8))): d[int(rng.integers(0,len(d)))]=int(rng.integers(0,256))
        else:
            lines=bytes(d).split(b"\n"); i=int(rng.integers(0,len(lines))); lines[i]=rng.choice([b"-1",b"99999999",b"nan",b"",b"abc def"]); d=bytearray(b"\n".join(lines))
        open(os.path.join(out,f"{seed}.pts"),"wb").write(bytes(d)); continue
    lines=str(rng.choice(texts)).split("\n")
    for _ in range(int(rng.integers(1,6))): lines.insert(int(rng.integers(0,len(lines))), str(rng.choice(tokens)))
    if rng.uniform()<0.2: lines=[l for l in lines if not l.startswith("vn")]
    if rng.uniform()<0.2: lines=[l for l in lines if not l.startswith("vt")]
    text="\n".join(lines)
    if rng.uniform()<0.15: text=text[:int(rng.integers(0,len(text)))]
    open(os.path.join(out,f"{seed}.obj"),"wb").write(text.encode("latin-1"))
print("ok")
