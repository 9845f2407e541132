// experiment: how much of the host decode is range decoder + model, how much is the rebuild of the BWT (dc::decode)?
#include "../dark_amd/csrc/entropy.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace dk;
static double now_ms(){ return std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  FILE*f=fopen("dc_stream.bin","rb"); size_t n,m; uint32_t origin; uint32_t init[256];
  (void)!fread(&n,8,1,f); (void)!fread(&m,8,1,f); (void)!fread(&origin,4,1,f); (void)!fread(init,4,256,f);
  std::vector<uint32_t> d(m); std::vector<uint8_t> s(m); (void)!fread(d.data(),4,m,f); (void)!fread(s.data(),1,m,f); fclose(f);
  std::vector<uint8_t> out(2*n+4096); size_t len=0;
  DcStream st; st.n=n; st.init=init; st.dist=d.data(); st.sym=s.data(); st.m=m; st.origin=origin;
  int rc=encode_block_stream(0,st,out.data(),out.size(),&len,1);
  printf("encode rc=%d len=%zu\n",rc,len);
  std::vector<uint8_t> bwt(n); uint32_t o2; int single;
  for(int it=0;it<3;it++){
    double t0=now_ms();
    rc=decode_block_stream(0,out.data(),len,n,bwt.data(),&o2,&single);
    double ms=now_ms()-t0;
    printf("full decode rc=%d %.1f ms %.1f ns/dist\n",rc,ms,ms*1e6/m);
  }
  for(int it=0;it<3;it++){
    DarkModel model; Decoder dec(out.data(),len);
    double t0=now_ms();
    // header
    bool active=true; size_t i=0; uint32_t v; bool ok=true;
    while(i<0xFF){ ok&=model.decode(0,dec,v); size_t num=v+((i==0&&active)?0:1); if(active) for(size_t c=i;c<i+num&&c<0x100;++c) ok&=model.decode((uint8_t)c,dec,v); active=!active; i+=num; }
    size_t bad=0;
    for(size_t k=0;k<m;k++){ ok&=model.decode(s[k],dec,v); bad+= v!=d[k]; }
    double ms=now_ms()-t0;
    printf("coder+model decode only: ok=%d bad=%zu %.1f ms %.1f ns/dist\n",(int)ok,bad,ms,ms*1e6/m);
  }
  // rebuild only
  for(int it=0;it<3;it++){
    double t0=now_ms();
    size_t used=0;
    rc=dc_decode_array(init,d.data(),m,bwt.data(),n,&used);
    double ms=now_ms()-t0;
    printf("rebuild only rc=%d %.1f ms %.1f ns/dist\n",rc,ms,ms*1e6/m);
  }
}
