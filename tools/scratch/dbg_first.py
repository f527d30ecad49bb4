import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import vaex_amd.superagg as sa
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_first.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
def sub(a,lo,hi,bins):
    s=(a-lo)/(hi-lo); i=np.nan_to_num(s*bins).astype(int)+2; i[s>=1]=bins+2; i[s<0]=1; i[np.isnan(s)]=0; return i
for seed in range(6):
    rng=np.random.default_rng(seed); n=40000
    x,y=rng.normal(0,1.2,n),rng.normal(0,1.2,n)
    if seed % 2: x[rng.random(n) < 0.01] = np.nan
    v,o=t._column(rng,"float64",n),t._column(rng,"float64",n)
    cx,cy=sub(x,-2,2,6),sub(y,-2,2,5)
    cell=cx+cy*9
    ok=~np.isnan(v)&~np.isnan(o)
    for chunks in ([(0,n)], [(0,15000),(15000,15001),(15001,n)]):
        rv,rm,a=t._run(sa,x,y,v,o,None,"float64","float64",False,chunks)
        vals,masked,orders=a.raw_result()
        got=rv.ravel(order="F"); mk=rm.ravel(order="F"); oo=np.asarray(orders).ravel(order="F")
        bad=0
        for c in range(72):
            m=(cell==c)&ok
            if m.any():
                idx=np.nonzero(m)[0]; j=idx[np.argmin(o[idx])]
                if got[c]!=v[j]:
                    bad+=1
                    if bad<3:
                        w=np.nonzero(v==got[c])[0]
                        print("  cell",c,"rows",len(idx),"expected row",j,"order",o[j],"| got value of row",w,"order",o[w] if len(w) else None,"cell of that row",cell[w] if len(w) else None,"reported order",oo[c])
        print(seed, len(chunks), "bad cells", bad)
