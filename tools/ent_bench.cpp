// host entropy stage alone on the sample written by tools/make_dc_sample.py: six encodes (DK_ENTROPY_THREADS selects the form), five decodes
#include "../dark_amd/csrc/entropy.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdlib>
using namespace dk;
int main(int argc,char**argv){
  FILE*f=fopen("dc_stream.bin","rb"); size_t n,m; uint32_t origin; uint32_t init[256];
  fread(&n,8,1,f); fread(&m,8,1,f); fread(&origin,4,1,f); fread(init,4,256,f);
  std::vector<uint32_t> d(m); std::vector<uint8_t> s(m); fread(d.data(),4,m,f); fread(s.data(),1,m,f); fclose(f);
  std::vector<uint8_t> out(2*n+4096); size_t len=0;
  DcStream st; st.n=n; st.init=init; st.dist=d.data(); st.sym=s.data(); st.m=m; st.origin=origin;
  for(int it=0;it<6;it++){
    auto t0=std::chrono::steady_clock::now();
    int rc=encode_block_stream(0,st,out.data(),out.size(),&len);
    double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
    printf("rc=%d len=%zu  %.1f ms  %.1f ns/dist  %.1f MB/s\n",rc,len,ms,ms*1e6/m,n/ms/1e3);
  }
  std::vector<uint8_t> bwt(n); uint32_t o2; int single;
  for(int it=0;it<5;it++){
  auto t0=std::chrono::steady_clock::now();
  int rc=decode_block_stream(0,out.data(),len,n,bwt.data(),&o2,&single);
  double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
  printf("decode rc=%d %.1f ms %.1f ns/dist\n",rc,ms,ms*1e6/m);
  }
}
