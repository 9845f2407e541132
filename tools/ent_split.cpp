// experiment: record the (total/shift, from, to) events of the dark model, then time the coder alone over the recording
#include "../dark_amd/csrc/entropy.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace dk;
struct Ev { uint32_t from, to, total; };
int main(){
  FILE*f=fopen("dc_stream.bin","rb"); size_t n,m; uint32_t origin; uint32_t init[256];
  (void)!fread(&n,8,1,f); (void)!fread(&m,8,1,f); (void)!fread(&origin,4,1,f); (void)!fread(init,4,256,f);
  std::vector<uint32_t> d(m); std::vector<uint8_t> s(m); (void)!fread(d.data(),4,m,f); (void)!fread(s.data(),1,m,f); fclose(f);
  // generate plausible events: random binary events with p from a table + 1 table event per distance
  std::vector<Ev> ev; ev.reserve(m*5);
  uint32_t seed=1;
  for(size_t k=0;k<m;k++){
    unsigned l=bit_length(d[k]+1);
    seed=seed*1664525u+1013904223u; uint32_t tot=3000+(seed>>20)%6000; uint32_t lo=(seed>>8)%(tot-200); ev.push_back({lo,lo+100+(seed&63),tot});
    for(unsigned i=1;i<l;i++){ seed=seed*1664525u+1013904223u; uint32_t z=1500+(seed>>21)%1000; bool b=(d[k]+1)>>(l-i-1)&1; ev.push_back(b?Ev{z,4096,0}:Ev{0,z,0}); }
  }
  std::vector<uint8_t> out(2*n+4096);
  for(int it=0;it<3;it++){
    auto t0=std::chrono::steady_clock::now();
    Encoder e(out.data(),out.size());
    for(const Ev&x:ev){ if(x.total) e.put(x.total,x.from,x.to); else e.put_pow2(12,x.from,x.to); }
    e.finish();
    double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
    printf("coder only: events=%zu len=%zu %.1f ms  %.1f ns/dist %.2f ns/event\n",ev.size(),e.size(),ms,ms*1e6/m,ms*1e6/ev.size());
  }
}
