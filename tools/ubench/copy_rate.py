import numpy as np, time
from concurrent.futures import ThreadPoolExecutor
n=128
src=[(np.random.randint(0,255,(480,640,3),dtype=np.uint8), np.random.randint(0,60000,(480,640),dtype=np.uint16)) for _ in range(n)]
ib=np.empty((n,480,640,3),np.uint8); db=np.empty((n,480,640),np.uint16)
def one(k):
    ib[k]=src[k][0]; db[k]=src[k][1]
for rep in range(2):
    t=time.perf_counter()
    for r in range(5):
        for k in range(n): one(k)
    dt=time.perf_counter()-t
    print("1 thread: %.0f frames/s"%(5*n/dt))
for w in (2,4,8,16):
    with ThreadPoolExecutor(w) as ex:
        t=time.perf_counter()
        for r in range(5):
            list(ex.map(one, range(n)))
        dt=time.perf_counter()-t
        print("%d threads: %.0f frames/s"%(w, 5*n/dt))
