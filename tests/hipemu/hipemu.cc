// hipemu runtime: see tests/hipemu/hip/hip_runtime.h.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <mutex>

namespace hipemu {

uint3e threadIdx_, blockIdx_;
dim3 blockDim_, gridDim_;
unsigned char* dyn_smem = nullptr;

namespace {
const size_t kStack = 512 * 1024;
const int kMaxThreads = 1024, kMaxWaves = kMaxThreads / 64;

struct Fiber {
  ucontext_t ctx;
  bool done;
};
ucontext_t sched_ctx;
Fiber fibers[kMaxThreads];
char* stacks = nullptr;
int cur = 0, nthreads = 0, alive = 0;
unsigned long events = 0;
unsigned long launches = 0;
int wait_kind[kMaxThreads];
void (*g_tramp)(void*) = nullptr;
void* g_args = nullptr;

int blk_gen = 0, blk_arrived = 0;
int wave_gen[kMaxWaves], wave_arrived[kMaxWaves], wave_alive[kMaxWaves];
unsigned long long ballot_acc[kMaxWaves], ballot_res[kMaxWaves];
unsigned long long xchg[kMaxWaves][2][64];

void set_tid(int t) {
  threadIdx_.x = t % blockDim_.x;
  threadIdx_.y = (t / blockDim_.x) % blockDim_.y;
  threadIdx_.z = t / (blockDim_.x * blockDim_.y);
}
void yield() {
  int me = cur;
  swapcontext(&fibers[me].ctx, &sched_ctx);
  cur = me;
  set_tid(me);
}
void release_block() { blk_arrived = 0; blk_gen++; events++; }
void release_wave(int w) {
  ballot_res[w] = ballot_acc[w];
  ballot_acc[w] = 0;
  wave_arrived[w] = 0;
  wave_gen[w]++;
  events++;
}
void entry() {
  g_tramp(g_args);
  int me = cur;
  fibers[me].done = true;
  alive--;
  events++;
  int w = me / 64;
  wave_alive[w]--;
  if (alive > 0 && blk_arrived == alive) release_block();
  if (wave_alive[w] > 0 && wave_arrived[w] == wave_alive[w]) release_wave(w);
}
int wave_rendezvous(int w) {  // returns the generation that was completed
  int gen = wave_gen[w];
  wave_arrived[w]++;
  if (wave_arrived[w] == wave_alive[w]) {
    release_wave(w);
    return gen;
  }
  wait_kind[cur] = 2;
  while (wave_gen[w] == gen) yield();
  wait_kind[cur] = 0;
  return gen;
}
}  // namespace

int lane_id() { return cur & 63; }

void block_barrier() {
  int gen = blk_gen;
  blk_arrived++;
  if (blk_arrived == alive) {
    release_block();
    return;
  }
  wait_kind[cur] = 1;
  while (blk_gen == gen) yield();
  wait_kind[cur] = 0;
}
void wave_barrier() { wave_rendezvous(cur / 64); }
// a fiber that polls memory another wavefront of its block writes (multi-wavefront k_lsd_grow_mw): give the others a turn.
// Not an event: a block whose fibers all poll is a deadlock and is reported as one.
void spin_yield() {
  wait_kind[cur] = 3;
  yield();
  wait_kind[cur] = 0;
}
unsigned long long wave_ballot(int pred) {
  int w = cur / 64;
  if (pred) ballot_acc[w] |= 1ull << (cur & 63);
  wave_rendezvous(w);
  return ballot_res[w];
}
unsigned long long wave_exchange(unsigned long long v, int src_lane, int) {
  int w = cur / 64;
  int par = wave_gen[w] & 1;
  xchg[w][par][cur & 63] = v;
  wave_rendezvous(w);
  return xchg[w][par][src_lane & 63];
}

void run_grid(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* args) {
  // one launch at a time: the fiber scheduler is global state, and a host may drive two handles from two threads (the
  // reference's Frame constructor runs ExtractORB and ExtractLSD concurrently, Frame.cc:224-227)
  static std::mutex launch_mutex;
  std::lock_guard<std::mutex> lock(launch_mutex);
  int n = (int)(block.x * block.y * block.z);
  if (n > kMaxThreads || n <= 0) {
    fprintf(stderr, "hipemu: bad block size %d\n", n);
    abort();
  }
  if (!stacks) {
    stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) abort();
  }
  std::vector<unsigned char> smem(shmem + 64);
  dyn_smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
  g_tramp = tramp;
  launches++;
  g_args = args;
  blockDim_ = block;
  gridDim_ = grid;
  nthreads = n;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx_.x = bx; blockIdx_.y = by; blockIdx_.z = bz;
        alive = n;
        blk_gen = 0; blk_arrived = 0;
        int nw = (n + 63) / 64;
        for (int w = 0; w < nw; w++) {
          wave_gen[w] = 0; wave_arrived[w] = 0; ballot_acc[w] = 0;
          wave_alive[w] = std::min(64, n - w * 64);
        }
        for (int t = 0; t < n; t++) {
          fibers[t].done = false;
          getcontext(&fibers[t].ctx);
          fibers[t].ctx.uc_stack.ss_sp = stacks + (size_t)t * kStack;
          fibers[t].ctx.uc_stack.ss_size = kStack;
          fibers[t].ctx.uc_link = &sched_ctx;
          makecontext(&fibers[t].ctx, (void (*)())entry, 0);
        }
        while (alive > 0) {
          unsigned long before = events;
          for (int t = 0; t < n; t++) {
            if (fibers[t].done) continue;
            cur = t;
            set_tid(t);
            swapcontext(&sched_ctx, &fibers[t].ctx);
          }
          if (alive > 0 && events == before) {
            fprintf(stderr, "hipemu: launch #%lu block dim %u grid %u,%u\n", launches, blockDim_.x, gridDim_.x, gridDim_.y);
            for (int t = 0; t < n && t < 64; t++) fprintf(stderr, "%d", fibers[t].done ? 9 : wait_kind[t]);
            fprintf(stderr, "\n");
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d threads alive, barrier arrivals %d\n", bx, by, bz, alive, blk_arrived);
            abort();
          }
        }
      }
  dyn_smem = nullptr;
}

}  // namespace hipemu
