// TEST INFRASTRUCTURE — host-vs-device differential test of the traversal.
//
// traversal.h is host/device code, so the same functions can run on the CPU and
// on the GPU over the same committed scene and the same rays.  Everything up to
// the raw hit record (instance, primitive, barycentrics / object-space point,
// distance, inside flag) is plain IEEE arithmetic and must agree bit for bit;
// the reconstructed surface may differ in the last ulps where libm functions
// (acosf, atan2f, sinf, cosf) are involved.  This test caught a hipcc -O2/-O3
// code-generation problem that silently dropped one store in the disk path.
//
// Built and run on the GPU box by tests/test_gpu_units.py:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off host_vs_device_walk.hip commit.cpp
//   ./a.out scene.mcsd  ->  "walk mismatch raw R surface S hits H of N"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include "path_core.h"
#include "host/commit.hpp"
using namespace mcpt;
struct In { V3 org, dir; };
struct Out { int hit; HitRaw h; float t; Surface s; };
MCPT_HD Out compute(const DeviceScene& sc, const In& in) {
  Out r; r.h.a = 111.f; r.h.b = 222.f; r.h.c = 333.f; r.h.prim = 7; r.h.inst = 9; r.h.inside = false;
  Ray ray = make_ray(in.org, in.dir); uint32_t rng = 1; TraceStats ts{0,0};
  r.hit = walk_scene<false,true,true,false>(sc, ray, rng, r.h, ts);
  r.t = ray.t_max;
  if (r.hit) r.s = make_surface<true,true>(sc, ray, r.h); else memset(&r.s,0,sizeof(r.s));
  return r;
}
__global__ void k(DeviceScene sc, const In* in, Out* out, int n) { int i = blockIdx.x*blockDim.x+threadIdx.x; if (i<n) out[i] = compute(sc, in[i]); }
template <class T> const T* up(const std::vector<T>& v){ T* d; hipMalloc(&d, v.size()*sizeof(T)+16); hipMemcpy(d, v.data(), v.size()*sizeof(T), hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
  FlatScene f = CommitScene(mcsd::Load(argv[1]));
  DeviceScene hs = f.HostView(); DeviceScene ds = hs;
  ds.nodes=up(f.nodes); ds.node_area=up(f.node_area); ds.tri_pos=up(f.tri_pos); ds.tri_attr=up(f.tri_attr); ds.instances=up(f.instances); ds.analytic=up(f.analytic);
  ds.light_inst=up(f.light_inst); ds.light_cdf=up(f.light_cdf); ds.textures=up(f.textures); ds.texels=up(f.texels); ds.bsdfs=up(f.bsdfs); ds.media=up(f.media); ds.emitters=up(f.emitters); ds.env_tables=up(f.env_tables); ds.lut_brdf=up(f.lut_brdf); ds.lut_albedo=up(f.lut_albedo);
  const int n = 50000;
  std::vector<In> in(n); std::mt19937 g(1); std::uniform_real_distribution<float> u(-1,1);
  for (auto& x : in) { x.org = {u(g)*1.5f, 2.0f+u(g), u(g)*1.5f}; V3 tgt = {u(g)*0.3f, 0.7f+u(g)*0.3f, u(g)*0.3f}; x.dir = normalize(tgt - x.org); }
  In* din; Out* dout; hipMalloc(&din, n*sizeof(In)); hipMalloc(&dout, n*sizeof(Out));
  hipMemcpy(din, in.data(), n*sizeof(In), hipMemcpyHostToDevice);
  k<<<(n+255)/256,256>>>(ds,din,dout,n); std::vector<Out> out(n); hipMemcpy(out.data(), dout, n*sizeof(Out), hipMemcpyDeviceToHost);
  int bad_raw=0, bad_surf=0, hits=0, shown=0;
  for (int i=0;i<n;++i){ Out h = compute(hs, in[i]); hits += h.hit;
    bool b = h.hit!=out[i].hit || (h.hit && (h.h.a!=out[i].h.a || h.h.b!=out[i].h.b || h.h.c!=out[i].h.c || h.h.prim!=out[i].h.prim || h.h.inst != out[i].h.inst || h.h.inside!=out[i].h.inside)) || h.t!=out[i].t;
    bool sb = h.hit && memcmp(&h.s.uv, &out[i].s.uv, sizeof(V2)+4*sizeof(V3))!=0;
    bad_raw += b; bad_surf += sb;
    if ((b||sb) && shown++<6) printf("%d hit %d/%d inst %u/%u a %.9g/%.9g b %.9g/%.9g c %.9g/%.9g t %.9g/%.9g uv %.9g,%.9g / %.9g,%.9g n.x %.9g/%.9g\n", i,h.hit,out[i].hit,h.h.inst,out[i].h.inst,h.h.a,out[i].h.a,h.h.b,out[i].h.b,h.h.c,out[i].h.c,h.t,out[i].t,h.s.uv.u,h.s.uv.v,out[i].s.uv.u,out[i].s.uv.v,h.s.normal.x,out[i].s.normal.x); }
  printf("walk mismatch raw %d surface %d hits %d of %d\n", bad_raw, bad_surf, hits, n);
}
