#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
// blocked k-ordered chain: blocks of BLK[l] k-values; chain inside block starts at 0 (first block: bias), folded sequentially
static void lin(const float* x,const float* W,const float* b,int K,int N,float* y,int blk){
  for(int n=0;n<N;++n){const float* w=W+(size_t)n*K; float tot=b[n]; 
    if(blk<=0){ float acc=tot; for(int k=0;k<K;++k) acc=fmaf(w[k],x[k],acc); y[n]=acc; continue;}
    int first=1; float sum=0;
    for(int k0=0;k0<K;k0+=blk){ float acc= first? b[n]:0.0f; int k1=k0+blk<K?k0+blk:K; for(int k=k0;k<k1;++k) acc=fmaf(w[k],x[k],acc);
      if(first){sum=acc;first=0;} else sum+=acc; }
    y[n]=sum; }
}
void fwd(const float* x,int64_t B,int F,const float* const* ew,const float* const* eb,const float* const* hw,const float* const* hb,const int* blk,float* out){
  static const int EN[7]={0,1024,512,256,128,64,9}; static const int HN[6]={3,128,256,128,64,1};
#pragma omp parallel
  { float* a=malloc(4*2048); float* c=malloc(4*2048);
#pragma omp for schedule(static)
  for(int64_t r=0;r<B;++r){ memcpy(a,x+r*F,4*F); int K=F;
    for(int l=0;l<6;++l){int N=EN[l+1]; lin(a,ew[l],eb[l],K,N,c,blk[l]); for(int n=0;n<N;++n) a[n]=(l<4)?(c[n]<0?0:c[n]):(l==4?tanhf(c[n]):c[n]); K=N;}
    float lat[9]; memcpy(lat,a,36);
    for(int g=0;g<3;++g){ memcpy(a,lat+3*g,12); int Kh=3; for(int l=0;l<5;++l){int N=HN[l+1]; lin(a,hw[g*5+l],hb[g*5+l],Kh,N,c,blk[6+l]); for(int n=0;n<N;++n) a[n]=(l<4)?(c[n]<0?0:c[n]):c[n]; Kh=N;} out[r*3+g]=a[0]; }
  } free(a);free(c);} }
