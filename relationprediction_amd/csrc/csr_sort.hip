// Hand-written stable sort of (small key, index) pairs: the "counting sort by (relation, dst)" behind every CSR the
// library builds -- the incidence CSR and the per-relation message list of the fed graph
// (code/extras/graph_representations.py:21-27 splits graph_edges; the reference then leaves the grouping to TF's
// SparseSoftmax / SparseTensorDenseMatMul / UnsortedSegmentSum kernels), and the entity / relation CSRs of the decoder
// batch (code/decoders/bilinear_diag.py:18-25 gathers).
//
// Keys are vertex or relation ids, or (vertex, directed relation) pairs (< 2^31), values the positions 0..n-1, so a
// least-significant-digit radix sort with 8-bit digits needs 1-4 passes.  Everything is a deterministic function of the input (no float or order-
// dependent atomics), which is what makes the fp32 sums downstream bitwise reproducible.
//
// One pass = two launches, for up to two independent sorts at once (grid.y picks the job):
//   k_sort_hist     block b counts the digit values of its 2048 items in LDS  -> table[b][256]          (u16)
//   k_sort_scatter  block b: digit base = exclusive scan over digits of the table's column sums, plus the column
//                   prefix over the blocks before b (every block reads the whole table: 512 bytes per block of input,
//                   cheaper than a scan launch -- up to kPrefixMinBlocks blocks, i.e. 1,048,576 items; beyond that the table
//                   traffic would grow quadratically (50 GB per pass at 20 M items), so k_sort_prefix first turns the
//                   table into per-block column prefixes, 256 blocks per workgroup, and a block reads its own row plus
//                   one total row per 256 blocks); stable rank inside the block without sorting anything:
//                     - lane rank among the lower lanes of the wave with the same digit: 8 ballots,
//                     - the 32 wave-rounds of the block (2 rounds x 16 waves, in item order) leave their per-digit
//                       counts in an LDS table [32][256] (u8), prefixed per digit;
//                   slot = digit base + blocks before + wave-rounds before + lane rank.
// The last pass can also write the inverse permutation (pos[value] = slot).
#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int kSortThreads = 1024;
constexpr int kSortItems = 2048;     // per block: 2 rounds of 1024
constexpr int kRounds = (kSortItems / kSortThreads) * (kSortThreads / 64);   // wave-rounds per block: 32
constexpr int kPrefixBlocks = 256;   // blocks per prefix group
constexpr int kPrefixMinBlocks = 512; // sorts of more blocks than this (1 M items) run k_sort_prefix: below, reading the
                                      // whole L2-resident table per block (<= 134 MB per pass) is cheaper than one more launch

struct SortJob {
  const uint32_t* key_in;
  const int32_t* val_in;      // nullptr: values are the positions 0..n-1
  uint32_t* key_out;
  int32_t* val_out;
  int32_t* pos_out;           // optional inverse permutation, written by this pass
  uint16_t* table;            // [nblocks][256]
  uint32_t* before;           // large sorts: [nblocks][256] column prefix inside the block's group of kPrefixBlocks blocks
  uint32_t* group_total;      // large sorts: [ngroups][256] column sums of every group
  int32_t n;
  int32_t shift;
};
struct SortJobs {
  SortJob j[2];
};

__global__ void __launch_bounds__(kSortThreads) k_sort_hist(SortJobs jobs) {
  const SortJob job = jobs.j[blockIdx.y];
  const int base = blockIdx.x * kSortItems;
  if (base >= job.n) return;
  __shared__ uint32_t hist[256];
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortItems / kSortThreads; ++r) {
    const int i = base + r * kSortThreads + threadIdx.x;
    if (i < job.n) atomicAdd(&hist[(job.key_in[i] >> job.shift) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256) job.table[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint16_t)hist[threadIdx.x];
}

// large sorts only: group g = blocks [256 g, 256 g + 256); thread d walks the group's column d
__global__ void __launch_bounds__(256) k_sort_prefix(SortJobs jobs) {
  const SortJob job = jobs.j[blockIdx.y];
  const int nblocks = (job.n + kSortItems - 1) / kSortItems;
  const int b0 = blockIdx.x * kPrefixBlocks;
  if (job.before == nullptr || b0 >= nblocks) return;
  const int b1 = min(nblocks, b0 + kPrefixBlocks), d = threadIdx.x;
  uint32_t run = 0;
  for (int b = b0; b < b1; ++b) {
    job.before[(size_t)b * 256 + d] = run;
    run += job.table[(size_t)b * 256 + d];
  }
  job.group_total[(size_t)blockIdx.x * 256 + d] = run;
}

__global__ void __launch_bounds__(kSortThreads) k_sort_scatter(SortJobs jobs) {
  const SortJob job = jobs.j[blockIdx.y];
  const int base = blockIdx.x * kSortItems;
  if (base >= job.n) return;
  const int nblocks = (job.n + kSortItems - 1) / kSortItems;
  __shared__ uint32_t part_total[4][256], part_before[4][256];
  __shared__ uint32_t digit_base[256];
  __shared__ uint8_t wcnt[kRounds][256];
  __shared__ uint16_t wpre[kRounds][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // column sums of the table: all blocks (total) and the blocks before this one, four threads per digit
  {
    const int d = tid & 255, part = tid >> 8;
    uint32_t tot = 0, bef = 0;
    if (job.before == nullptr) {
      for (int b = part; b < nblocks; b += 4) {
        const uint32_t v = job.table[(size_t)b * 256 + d];
        tot += v;
        bef += b < (int)blockIdx.x ? v : 0u;
      }
    } else {
      const int ngroups = (nblocks + kPrefixBlocks - 1) / kPrefixBlocks, mine = (int)blockIdx.x / kPrefixBlocks;
      for (int g = part; g < ngroups; g += 4) {
        const uint32_t v = job.group_total[(size_t)g * 256 + d];
        tot += v;
        bef += g < mine ? v : 0u;
      }
      if (part == 0) bef += job.before[(size_t)blockIdx.x * 256 + d];
    }
    part_total[part][d] = tot;
    part_before[part][d] = bef;
  }
  for (int i = tid; i < kRounds * 256; i += kSortThreads) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  if (tid < 256) {
    // exclusive scan over the 256 digit totals by the first four waves
    const uint32_t tot = part_total[0][tid] + part_total[1][tid] + part_total[2][tid] + part_total[3][tid];
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    part_total[0][tid] = incl;                 // inclusive within the wave (own slot only: no hazard)
    part_total[1][tid] = tot;
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t before_waves = 0;
    for (int w = 0; w < wave; ++w) before_waves += part_total[0][w * 64 + 63];
    digit_base[tid] = before_waves + part_total[0][tid] - part_total[1][tid] + part_before[0][tid] +
                      part_before[1][tid] + part_before[2][tid] + part_before[3][tid];
  }

  // the block's items, in item order: round r, wave w  ->  wave-round q = 16 r + w
  uint32_t key[kSortItems / kSortThreads];
  int32_t val[kSortItems / kSortThreads];
  int rank[kSortItems / kSortThreads];
#pragma unroll
  for (int r = 0; r < kSortItems / kSortThreads; ++r) {
    const int i = base + r * kSortThreads + tid;
    const bool valid = i < job.n;
    key[r] = valid ? job.key_in[i] : 0u;
    val[r] = valid ? (job.val_in ? job.val_in[i] : i) : 0;
    const uint32_t digit = (key[r] >> job.shift) & 255u;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (digit >> bit) & 1u;
      const unsigned long long vote = __ballot(one);
      same &= one ? vote : ~vote;
    }
    rank[r] = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank[r] == 0) wcnt[r * (kSortThreads / 64) + wave][digit] = (uint8_t)__popcll(same);
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t run = 0;
#pragma unroll 8
    for (int q = 0; q < kRounds; ++q) {
      wpre[q][tid] = (uint16_t)run;
      run += wcnt[q][tid];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortItems / kSortThreads; ++r) {
    const int i = base + r * kSortThreads + tid;
    if (i < job.n) {
      const uint32_t digit = (key[r] >> job.shift) & 255u;
      const uint32_t slot = digit_base[digit] + wpre[r * (kSortThreads / 64) + wave][digit] + (uint32_t)rank[r];
      job.key_out[slot] = key[r];
      job.val_out[slot] = val[r];
      if (job.pos_out) job.pos_out[val[r]] = (int32_t)slot;
    }
  }
}

int passes_for(uint32_t max_key) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) <= max_key) ++bits;
  return (bits + 7) / 8;
}

}  // namespace

// uint16 elements: the [nblocks][256] count table, and for large sorts behind it the uint32 prefix rows [nblocks][256]
// and group totals [ngroups][256]
size_t sort_table_elems(size_t n) {
  const size_t nblocks = (n + kSortItems - 1) / kSortItems + 1;
  size_t elems = nblocks * 256;
  if (nblocks > (size_t)kPrefixMinBlocks) elems += 2 * (nblocks + nblocks / kPrefixBlocks + 2) * 256;
  return elems;
}

// Stable sort of up to two independent (key, position) arrays by key, on the context's current stream.
// keys < 2^31 (max_key bounds the passes); key_tmp / val_tmp: scratch of n elements; the result lands in key_out /
// val_out (val = original positions, in key order, ties in position order); pos_out (nullable) = its inverse.
rgcn_status sort_pairs(rgcn_ctx* c, const char* tag, int njobs, const SortSpec* specs) {
  if (njobs < 1 || njobs > 2) RGCN_FAIL(c, RGCN_ERR_INVALID, "internal: sort_pairs takes one or two jobs");
  int passes[2] = {0, 0}, max_passes = 0, max_blocks = 0;
  for (int k = 0; k < njobs; ++k) {
    if (specs[k].n < 0 || specs[k].max_key >= (1u << 31)) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "internal: sort key range");
    passes[k] = specs[k].n > 0 ? passes_for(specs[k].max_key) : 0;
    if (passes[k] > max_passes) max_passes = passes[k];
    const int nb = (int)((specs[k].n + kSortItems - 1) / kSortItems);
    if (nb > max_blocks) max_blocks = nb;
  }
  if (max_passes == 0) return RGCN_OK;
  double bytes = 0;
  for (int k = 0; k < njobs; ++k) bytes += 16.0 * specs[k].n * passes[k];
  ProfScope ps(c, tag, bytes, 0);
  for (int p = 0; p < max_passes; ++p) {
    SortJobs jobs;
    bool large = false;
    for (int k = 0; k < 2; ++k) {
      SortJob& j = jobs.j[k];
      if (k >= njobs || p >= passes[k]) {      // nothing (left) to do for this job: an empty job returns at once
        j = SortJob{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
        continue;
      }
      const SortSpec& s = specs[k];
      // ping-pong so that the LAST pass writes key_out / val_out
      const bool to_out = ((passes[k] - p) % 2) == 1;
      j.key_in = p == 0 ? s.key_in : (to_out ? s.key_tmp : s.key_out);
      j.val_in = p == 0 ? nullptr : (to_out ? s.val_tmp : s.val_out);
      j.key_out = to_out ? s.key_out : s.key_tmp;
      j.val_out = to_out ? s.val_out : s.val_tmp;
      j.pos_out = p == passes[k] - 1 ? s.pos_out : nullptr;
      j.table = s.table;
      const size_t nbk = (size_t)((s.n + kSortItems - 1) / kSortItems);
      j.before = nullptr;
      j.group_total = nullptr;
      if (nbk > (size_t)kPrefixMinBlocks) {
        j.before = reinterpret_cast<uint32_t*>(s.table + (nbk + 1) * 256);
        j.group_total = j.before + (nbk + 1) * 256;
        large = true;
      }
      j.n = (int32_t)s.n;
      j.shift = 8 * p;
    }
    hipLaunchKernelGGL(k_sort_hist, dim3(max_blocks, njobs), dim3(kSortThreads), 0, c->stream, jobs);
    if (large)
      hipLaunchKernelGGL(k_sort_prefix, dim3((max_blocks + kPrefixBlocks - 1) / kPrefixBlocks, njobs), dim3(256), 0,
                         c->stream, jobs);
    hipLaunchKernelGGL(k_sort_scatter, dim3(max_blocks, njobs), dim3(kSortThreads), 0, c->stream, jobs);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

}  // namespace rgcn
