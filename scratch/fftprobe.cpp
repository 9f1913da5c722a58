#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
int try_plan(int nx,int ny,int nz,bool inverse,bool padded){
  size_t lengths[2]={(size_t)nx,(size_t)ny};
  int nkx=nx/2+1; size_t sy=nx, sz=padded? (size_t)nx*(ny+4):(size_t)nx*ny;
  size_t rstr[2]={1,sy}, cstr[2]={1,(size_t)nkx}, off[1]={0};
  rocfft_plan_description d; rocfft_plan_description_create(&d);
  rocfft_status s;
  if(!inverse) s=rocfft_plan_description_set_data_layout(d,rocfft_array_type_real,rocfft_array_type_hermitian_interleaved,off,off,2,rstr,sz,2,cstr,(size_t)nkx*ny);
  else s=rocfft_plan_description_set_data_layout(d,rocfft_array_type_hermitian_interleaved,rocfft_array_type_real,off,off,2,cstr,(size_t)nkx*ny,2,rstr,sz);
  rocfft_plan p=nullptr;
  rocfft_status s2=rocfft_plan_create(&p,rocfft_placement_notinplace,inverse?rocfft_transform_type_real_inverse:rocfft_transform_type_real_forward,rocfft_precision_double,2,lengths,nz,d);
  printf("nx=%d ny=%d nz=%d inv=%d padded=%d layout=%d create=%d\n",nx,ny,nz,inverse,padded,(int)s,(int)s2);
  if(p) rocfft_plan_destroy(p);
  rocfft_plan_description_destroy(d);
  return s2;
}
int main(){
  rocfft_setup();
  int sizes[][3]={{8,8,6},{12,8,6},{8,12,10},{4,4,3},{16,8,12},{16,16,8},{20,12,10},{128,8,6},{32,16,16},{64,48,40},{256,256,4}};
  for(auto&s:sizes) for(int inv=0;inv<2;++inv) for(int pad=0;pad<2;++pad) try_plan(s[0],s[1],s[2],inv,pad);
  rocfft_cleanup();
}
