"""The host triangulator (csrc/amt_delaunay.hip is plain C++ for the host) under AddressSanitizer + UBSan ON THE CPU (the GPU pool
allows no sanitizer runs): the reference's fixtures, an exact lattice (ties everywhere), duplicates, collinear and tiny inputs, NaN
targets, random clouds at several scales — through the sequential build and through the parallel one (strips on threads, joined
at their seams).  usage (build container):
  g++ -O1 -g -std=c++17 -pthread -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -DAMT_DELAUNAY_STANDALONE -x c++ -o /tmp/libdel_asan.so auromat_amd/csrc/amt_delaunay.hip
  LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_delaunay.py
and under ThreadSanitizer (the strips, the copy into one structure and the compaction run on threads):
  g++ -O1 -g -std=c++17 -pthread -fPIC -shared -fsanitize=thread -DAMT_DELAUNAY_STANDALONE -x c++ -o /tmp/libdel_asan.so auromat_amd/csrc/amt_delaunay.hip
  LD_PRELOAD=$(g++ -print-file-name=libtsan.so) python tools/asan_delaunay.py"""
import ctypes as C, numpy as np, sys
lib=C.CDLL('/tmp/libdel_asan.so')
THREADS=[1]
def run(pts, targets):
    pts=np.ascontiguousarray(pts,dtype=np.float64)
    h=C.c_void_p()
    rc=lib.amt_delaunay_create_threads(pts.ctypes.data_as(C.c_void_p), C.c_int64(len(pts)), C.c_int32(THREADS[0]), C.c_int64(300), C.byref(h))
    if rc!=0: return rc
    nt,nn,nd=C.c_int64(),C.c_int64(),C.c_int64()
    lib.amt_delaunay_sizes(h,C.byref(nt),C.byref(nn),C.byref(nd))
    tri=np.empty((nt.value,3),np.int32); nbr=np.empty((nt.value,3),np.int32)
    lib.amt_delaunay_triangles(h,tri.ctypes.data_as(C.c_void_p),nbr.ctypes.data_as(C.c_void_p))
    indptr=np.empty(len(pts)+1,np.int64); ind=np.empty(nn.value,np.int32)
    lib.amt_delaunay_vertex_neighbours(h,indptr.ctypes.data_as(C.c_void_p),ind.ctypes.data_as(C.c_void_p))
    st=(C.c_int64*4)(); lib.amt_delaunay_stats(h,st)
    m=len(targets); tg=np.ascontiguousarray(targets,dtype=np.float64)
    v=np.empty((m,3),np.int32); c=np.empty((m,3,2)); hn=np.empty((m,3),np.uint8)
    lib.amt_delaunay_locate(h,tg.ctypes.data_as(C.c_void_p),C.c_int64(m),v.ctypes.data_as(C.c_void_p),c.ctypes.data_as(C.c_void_p),hn.ctypes.data_as(C.c_void_p))
    lib.amt_delaunay_destroy(h)
    return nt.value, nd.value, list(st)
def cases():
  rng=np.random.RandomState(0)
  for name in ('resample_nearest_iss029','resample_nearest_synth_pole'):
      z=np.load('/root/repo/tests/golden/%s.npz'%name); la,lo=z['lats_c'],z['lons_c']; ok=~np.isnan(la.ravel())
      pts=np.column_stack((la.ravel()[ok],lo.ravel()[ok]))
      tg=np.column_stack((rng.uniform(pts[:,0].min()-1,pts[:,0].max()+1,2000),rng.uniform(pts[:,1].min()-1,pts[:,1].max()+1,2000)))
      print(name, run(pts,tg))
  # degenerate inputs: exact lattice, duplicates, collinear, tiny, NaN targets, random clouds
  g=np.column_stack([a.ravel() for a in np.mgrid[0:40,0:30].astype(float)])
  print('lattice',run(g, np.array([[1.5,2.5],[np.nan,0],[100,100],[0,0],[39,29]])))
  print('dups',run(np.vstack((g[:100],g[:50])), g[:10]+0.25))
  print('collinear',run(np.column_stack((np.arange(50.),np.arange(50.)*3)), g[:3]))
  print('three',run(np.array([[0,0],[1,0],[0,1.]]), np.array([[0.2,0.2],[2,2.]])))
  for k in range(20):
      n=rng.randint(3,3000)
      pts=rng.rand(n,2)*rng.choice([1e-3,1,1e5])+rng.choice([0,1e6])
      r=run(pts, rng.rand(200,2)*2-0.5)
  print('random clouds ok', r)
  for k in range(6):
    pts=rng.rand(rng.randint(5000,40000),2)*[rng.choice([0.1,1,30]),1.0]
    r=run(pts, rng.rand(200,2))
  print('larger clouds ok', r)
for t in (1, 2, 5):
  THREADS[0]=t
  print('== threads', t)
  cases()
