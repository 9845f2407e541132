// experiment: cost of the two halves of the two-thread encoder when each runs alone (no hand-off): the dark model writing its
// 16-bit decision units into memory, and the range coder reading them back.
#include "../dark_amd/csrc/entropy.cpp"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace dk;
struct VecSink {
    std::vector<uint16_t> u;
    bool put(uint32_t total, uint32_t from, uint32_t to) { u.push_back(from); u.push_back(to); u.push_back(total); return true; }
    bool put_pow2(unsigned shift, uint32_t from, uint32_t to) { return put(1u << shift, from, to); }
    bool put_bit12(uint32_t zero, bool one) { u.push_back(0x8000u | (one ? 0x4000u : 0u) | zero); return true; }
    bool finish() { return true; }
    int error() const { return 0; }
};
static double now_ms(){ return std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  FILE*f=fopen("dc_stream.bin","rb"); size_t n,m; uint32_t origin; uint32_t init[256];
  (void)!fread(&n,8,1,f); (void)!fread(&m,8,1,f); (void)!fread(&origin,4,1,f); (void)!fread(init,4,256,f);
  std::vector<uint32_t> d(m); std::vector<uint8_t> s(m); (void)!fread(d.data(),4,m,f); (void)!fread(s.data(),1,m,f); fclose(f);
  DcStream st; st.n=n; st.init=init; st.dist=d.data(); st.sym=s.data(); st.m=m; st.origin=origin;
  std::vector<uint8_t> out(2*n+4096);
  for (int it=0; it<3; ++it) {
    VecSink sink; sink.u.reserve(m*8);
    auto model = std::make_unique<DarkModel>();
    double t0=now_ms();
    int rc = write_stream(*model, st, sink);
    double t1=now_ms();
    // consumer alone
    RangeState rs; uint8_t *p = out.data();
    const uint16_t *ev = sink.u.data(); size_t cnt = sink.u.size(); size_t nev=0;
    for (size_t k=0;k<cnt;) {
        const uint32_t u=ev[k]; const uint32_t span=rs.hi-rs.low; int nb;
        if (u&0x8000u){ const uint32_t zero=u&0xFFFu, r=span>>12; const bool one=(u&0x4000u)!=0; nb=rs.narrow(r, one?zero:0u, one?4096u:zero, p); k+=1; }
        else { const uint32_t to=ev[k+1], total=ev[k+2]; const uint32_t r=span/total; nb=rs.narrow(r,u,to,p); k+=3; }
        p+=nb; ++nev;
    }
    double t2=now_ms();
    {   // uniform events: every decision as (from, to, 2^64 / total rounded up) -> no division, no shift/divide branch in the coder
        struct UEv { uint64_t inv; uint32_t from, to; };
        std::vector<UEv> ue; ue.reserve(nev);
        for (size_t k=0;k<cnt;) { const uint32_t u=ev[k];
            if (u&0x8000u){ const uint32_t zero=u&0xFFFu; const bool one=(u&0x4000u)!=0; ue.push_back({1ull<<52, one?zero:0u, one?4096u:zero}); k+=1; }
            else { const uint32_t total=ev[k+2]; ue.push_back({(total&(total-1))? (~0ull/total+1) : ((1ull<<63)/total*2), u, ev[k+1]}); k+=3; } }
        RangeState r2; uint8_t *q = out.data();
        double u0=now_ms();
        for (const UEv &x : ue) {
            const uint32_t span=r2.hi-r2.low;
            const uint32_t rr=static_cast<uint32_t>((static_cast<unsigned __int128>(span)*x.inv)>>64);
            q += r2.narrow(rr, x.from, x.to, q);
        }
        double u1=now_ms();
        printf("   uniform coder: %.1f ms %.2f ns/dist, out %zu bytes (%s)\n", u1-u0,(u1-u0)*1e6/m,(size_t)(q-out.data()), (size_t)(q-out.data())==(size_t)(p-out.data())?"same":"DIFFERENT");
    }
    {   // model halves alone
        struct ExpSide { DarkModel &m; bool encode(uint32_t dist, uint8_t sym, VecSink &e) { return m.encode_exponent(dist, sym, e); } };
        struct ManSide { DarkModel &m; bool encode(uint32_t dist, uint8_t, VecSink &e) { return m.encode_mantissa_modelled(dist, e); } };
        struct FlatSide { bool encode(uint32_t dist, uint8_t, VecSink &e) { return DarkModel::encode_mantissa_flat(dist, e); } };
        auto m1 = std::make_unique<DarkModel>(); auto m2 = std::make_unique<DarkModel>();
        VecSink s1, s2, s3; s1.u.reserve(m*4); s2.u.reserve(m*4); s3.u.reserve(m*8);
        ExpSide e1{*m1}; ManSide e2{*m2}; FlatSide e3;
        double a0=now_ms(); write_stream(e1, st, s1); double a1=now_ms(); write_stream(e2, st, s2); double a2=now_ms(); write_stream(e3, st, s3); double a3=now_ms();
        printf("   exponent half: %.2f ns/dist (%zu units) | modelled mantissa: %.2f ns/dist (%zu units) | flat mantissa: %.2f ns/dist (%zu units)\n",
               (a1-a0)*1e6/m, s1.u.size(), (a2-a1)*1e6/m, s2.u.size(), (a3-a2)*1e6/m, s3.u.size());
    }
    printf("rc=%d models+units: %.1f ms %.2f ns/dist | coder over %zu events (%zu units, %.1f MB): %.1f ms %.2f ns/dist, out %zu bytes\n", rc, t1-t0,(t1-t0)*1e6/m, nev, cnt, cnt*2/1e6, t2-t1,(t2-t1)*1e6/m, (size_t)(p-out.data()));
  }
}
