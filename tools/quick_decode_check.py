import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import crackle_amd
from crackle_amd import synth
from oracle import oracle
chk = oracle.best()
ok = True
for shape, dt, cell in [((320,288,5), np.uint32,(16,16,4)), ((1024,96,3), np.uint16,(32,32,8)), ((64,64,16),np.uint8,(8,8,4)), ((2048,40,2),np.uint32,(32,32,8)), ((36,300,3),np.uint64,(8,8,4)), ((4,4,2),np.uint8,(2,2,1)), ((1024,1024,4),np.uint32,(32,32,8))]:
  arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=31, cell=cell))
  for kw in (dict(), dict(markov_model_order=3), dict(allow_pins=True)):
    b = chk.compress(arr, **kw)
    for env in ({}, {"CKL_DECODE_RASTER":"1"}, {"CKL_REC_CAP":"3"}, {"CKL_LDS_CONTROLS":"64"}, {"CKL_RESOLVE_CAP":"7"}):
      for k,v in env.items(): os.environ[k]=v
      try:
        got = crackle_amd.decompress(b)
        good = np.array_equal(got, arr)
      except Exception as e:
        good = False; print("EXC", e)
      for k in env: del os.environ[k]
      print(shape, np.dtype(dt).name, kw, env, "OK" if good else "MISMATCH", flush=True)
      ok &= good
print("ALL OK" if ok else "FAILURES")
