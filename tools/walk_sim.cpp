// tools/walk_sim.cpp -- SIMT scheduling model of the stack walk (measurement tool, not product): records the per-segment node
// sequences of the CPU build of the core on a reference scene and replays them on 64-lane waves under scheduling policies
// (every kind per step / vote for one kind per step; with and without refill).  Build: g++ -O2 -std=c++17 -ffp-contract=off
// -Iinclude -Iraytracing-1w_amd/csrc tools/walk_sim.cpp -o /tmp/walk_sim -Lraytracing-1w_amd -lrt1w -Wl,-rpath,$PWD/raytracing-1w_amd ; /tmp/walk_sim <arm> <W> <H> <spp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cstdint>
#include <algorithm>
static thread_local std::vector<uint8_t>* g_seq = nullptr;
#define RT_STAT_VISIT(kind) do { if (g_seq) g_seq->push_back((uint8_t)(kind)); } while (0)
#include "rt_core.h"
#include "rt1w.h"
struct HostStack { uint32_t e[64]; int sp=0; void push(uint32_t v){e[sp++]=v;} uint32_t pop(){return e[--sp];} };
struct cam_bg { RtCamera cam; RtV3 bg; uint32_t root, pad; };
// op classes: 0 box, 1 sphere, 2 msphere, 3 rect, 4 wrap, 5 medium
static int cls(uint8_t k){ return k<=1?0 : k==2?1 : k==3?2 : k<=6?3 : k<=9?4 : 5; }
static const double COST[6]={1.0,3.0,3.8,1.3,0.8,1.0};
int main(int argc,char**argv){
  int arm=atoi(argv[1]); int W=atoi(argv[2]),H=atoi(argv[3]),spp=atoi(argv[4]);
  std::vector<uint8_t> earth(1024*512*3,128);
  rt1w_scene* s=nullptr; uint32_t def[3];
  if(rt1w_scene_build_reference(arm,1,(double)W/H,earth.data(),1024,512,&s,def)){printf("fail\n");return 1;}
  std::vector<std::vector<uint8_t>> a(7);
  for(int i=0;i<7;i++){int64_t n=rt1w_scene_copy_flat(s,i,nullptr,0); a[i].resize(n>0?n:16); rt1w_scene_copy_flat(s,i,a[i].data(),a[i].size());}
  rt1w_scene_info inf; rt1w_scene_get_info(s,&inf);
  RtSceneView sc; memset(&sc,0,sizeof sc);
  const cam_bg* cb=(const cam_bg*)a[6].data();
  sc.nodes=(const RtNode*)a[0].data(); sc.lights=(const RtNode*)a[1].data(); sc.materials=(const RtMaterial*)a[2].data(); sc.textures=(const RtTexture*)a[3].data();
  sc.perlin=(const RtPerlin*)a[4].data(); sc.images=a[5].data(); sc.root=cb->root; sc.n_nodes=inf.n_nodes; sc.n_lights=inf.n_lights; sc.n_materials=inf.n_materials; sc.n_textures=inf.n_textures;
  sc.camera=cb->cam; sc.background=cb->bg;
  RtFrame f; memset(&f,0,sizeof f); f.width=W; f.height=H; f.tile_w=W; f.tile_h=H; f.spp=spp; f.max_depth=50; f.chunk=spp; f.n_chunks=1;
  HostStack stk; RtGlobalNodes ns{sc.nodes};
  // paths in 8x8 pixel-block order like the kernel; per bounce generation: queue order = path order (wavefront) 
  std::vector<std::vector<std::vector<uint8_t>>> bounce; // bounce[b][i] = seq
  std::vector<RtPath> paths;
  for(int by=0;by<H;by+=8)for(int bx=0;bx<W;bx+=8)for(int k=0;k<spp;k++)for(int y=by;y<by+8&&y<H;y++)for(int x=bx;x<bx+8&&x<W;x++){ RtPath p; rt_path_begin(sc,f,x,y,k,p); paths.push_back(p);}  
  std::vector<size_t> alive(paths.size()); for(size_t i=0;i<alive.size();i++)alive[i]=i;
  // also per-path sequences list for megakernel sim
  std::vector<std::vector<std::vector<uint8_t>>> per_path(paths.size());
  while(!alive.empty()){
    std::vector<std::vector<uint8_t>> segs; std::vector<size_t> next;
    for(size_t id: alive){ RtPath&p=paths[id]; if(p.depth_left==0){ rt_path_step<RtCfgV3>(sc,ns,p,stk); continue;} std::vector<uint8_t> q; g_seq=&q; rt_path_step<RtCfgV3>(sc,ns,p,stk); g_seq=nullptr; per_path[id].push_back(q); segs.push_back(std::move(q)); if(p.alive) next.push_back(id);}    
    bounce.push_back(std::move(segs)); alive.swap(next);
  }
  size_t nseg=0; double work=0; for(auto&b:bounce)for(auto&q:b){nseg++; for(auto k:q)work+=COST[cls(k)];}
  printf("arm %d nodes %u segments %zu, ideal lane-work per segment %.1f cost units\n",arm,inf.n_nodes,nseg,work/nseg);
  // --- wavefront trace sim with refill: lanes pull next segment of the bounce queue when done; policies
  auto sim=[&](int policy,bool refill,int thresh){ double total=0; double lanework=0; double steps=0;
    for(auto&b:bounce){ size_t nq=b.size(); size_t nextq=0; // waves process queue: each wave persistent; simulate a single wave stream consuming whole queue sequentially in chunks (many waves in parallel ~ same efficiency)
      // emulate W waves each taking interleaved batches of 64: simply simulate one wave over consecutive groups with refill
      const uint8_t* cur[64]; size_t len[64], pos[64]; bool have[64]; for(int l=0;l<64;l++)have[l]=false;
      for(;;){ // refill
        int idle=0; for(int l=0;l<64;l++) if(!have[l]) idle++;
        bool any=false; for(int l=0;l<64;l++) any|=have[l];
        if((refill && idle>=thresh) || !any){ for(int l=0;l<64;l++) if(!have[l]&&nextq<nq){ cur[l]=b[nextq].data(); len[l]=b[nextq].size(); pos[l]=0; have[l]=len[l]>0; nextq++; } any=false; for(int l=0;l<64;l++) any|=have[l]; if(!any){ if(nextq>=nq) break; else continue; } }
        int cnt[6]={0}; for(int l=0;l<64;l++) if(have[l]) cnt[cls(cur[l][pos[l]])]++; steps+=1;
        if(policy==0){ for(int c=0;c<6;c++) if(cnt[c]){ total+=COST[c]; lanework+=cnt[c]*COST[c]; } for(int l=0;l<64;l++) if(have[l]){ if(++pos[l]>=len[l]) have[l]=false; } }
        else { // vote: run the class with max lanes*... (policy1: max count; policy2: max count*cost efficiency = count)
          int best=-1; double bs=-1; for(int c=0;c<6;c++) if(cnt[c]){ double sc_= policy==1? cnt[c] : cnt[c]/ (1.0); if(policy==3){ // boxes unless leaf lanes >= thresh2 or no boxes
                } if(sc_>bs){bs=sc_;best=c;} }
          if(policy==3){ int leaf=cnt[1]+cnt[2]+cnt[3]; if(cnt[0]+cnt[4]+cnt[5]>0 && leaf<24) { // run non-leaf classes present
                for(int c: {0,4,5}) if(cnt[c]){ total+=COST[c]; lanework+=cnt[c]*COST[c]; } for(int l=0;l<64;l++) if(have[l]){ int c=cls(cur[l][pos[l]]); if(c==0||c>=4){ if(++pos[l]>=len[l]) have[l]=false; } } continue; }
              else { for(int c: {1,2,3}) if(cnt[c]){ total+=COST[c]; lanework+=cnt[c]*COST[c]; } for(int l=0;l<64;l++) if(have[l]){ int c=cls(cur[l][pos[l]]); if(c>=1&&c<=3){ if(++pos[l]>=len[l]) have[l]=false; } } continue; } }
          total+=COST[best]; lanework+=cnt[best]*COST[best]; for(int l=0;l<64;l++) if(have[l]&&cls(cur[l][pos[l]])==best){ if(++pos[l]>=len[l]) have[l]=false; } }
      }
    }
    printf("policy %d refill %d thresh %d: wave-cost per segment %.2f, utilization %.3f, wave-steps per 64 segments %.1f\n",policy,refill,thresh,total*64/ nseg /64, lanework/(total*64), steps*64/nseg); };
  sim(0,false,64); sim(0,true,8); sim(1,false,64); sim(1,true,32); sim(1,true,16); sim(1,true,8); sim(1,true,1);
}
