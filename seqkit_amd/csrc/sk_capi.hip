// sk_capi.hip — the C-ABI of include/seqkit_hip.h on top of the gfx950 kernels.
// No CPU fallback lives here: without a GPU sk_create() fails and every other call needs a ctx.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: librccl is loaded with dlopen on first use
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/seqkit_hip.h"
#include "sk_internal.h"

struct sk_ctx {
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;     // second lane of the host entry points' chunk pipeline (H2D of chunk i+1 under kernel / D2H of chunk i)
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_pipe = nullptr;
	std::string err;
	// barcode table
	bool have_table = false;
	int S = 0, L = 0, W = 0, max_diff = 1;
	uint8_t *d_raw = nullptr;
	uint32_t *d_onehot = nullptr;
	uint8_t *d_lut = nullptr;
	uint8_t *d_bs = nullptr;
	int bs_bytes = 0, bs_mm_off = 0, G = 0;
	uint8_t *d_nbr = nullptr;          // neighbourhood table of the sheet (sk_lut.h, demux_lut_kernel), when it has one: slots, then the ambiguity pairs
	sk::LutDev nbr{};
	std::vector<uint8_t> sheet;        // the sheet as given to sk_set_barcodes (the table is built from it on first use)
	bool nbr_tried = false;
	int64_t nbr_keys = 0, nbr_bytes = 0;
	int detail_mode = SK_DETAIL_FULL;
	unsigned long long *d_counts = nullptr;    // u64[S+3]
	unsigned long long *d_counts_wide = nullptr;   // TileArgs::counts_wide: 16 x (S+3) lines; folded into d_counts by fold_counts()
	bool wide_dirty = false;
	unsigned long long *d_count_rep = nullptr; // BarcodeDev::count_rep
	int count_rep_pitch = 0;
	size_t count_rep_set = 0;                  // u64 words of one set
	// workspace for the host-pointer entry points
	uint8_t *ws = nullptr;
	size_t ws_bytes = 0;
	// sk_fused_pass_many_dev: the batch descriptors of a many-batch launch travel through a ring of pinned / device slots
	sk::ManyBatch *many_pin = nullptr, *many_dev = nullptr;
	hipEvent_t many_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	int many_next = 0;
	// the sheet / pitch / output set for which launch_tile_pass last said "no one-launch form": the next many-batch call of that shape goes
	// straight to its batches' own launches (the refused attempt cost an event and the descriptors: 3 us a call)
	const void *many_no_tab = nullptr;
	int many_no_stride = 0, many_no_detail = -1;
	uint8_t *pin = nullptr;            // a pinned landing area (sk_bgzf_deflate: the compressed slots come back here)
	size_t pin_bytes = 0;
	// buffers that stay with the ctx from one call to the next (sk::ctx_keep: sk_bam_file_reduce's device and pinned buffers)
	struct Kept { void *p = nullptr; size_t cap = 0; bool pinned = false; } kept[8];
	void *ext = nullptr;               // an object another translation unit keeps with the ctx (sk_bamfile.cpp: its mapped output range), and how to free it
	void (*ext_free)(void *) = nullptr;
	sk::Census *census = nullptr;
	ncclComm_t comm = nullptr;         // one-process-per-GPU communicator (sk_comm_init_rank)
	int comm_ranks = 0;
	// `fasta gc content`: the genome, resident
	uint8_t *d_genome = nullptr;
	int64_t genome_len = 0;
	// `sam count` region tables
	int cnt_n_chr = 0;
	int64_t cnt_n_regions = 0;
	uint8_t *d_cnt = nullptr;          // one allocation: chr_off, rstart, rend, rpmax, ridx, counts
	int32_t *d_chr_off = nullptr, *d_ridx = nullptr;
	uint32_t *d_rstart = nullptr, *d_rend = nullptr, *d_rpmax = nullptr, *d_region_frags = nullptr;
};

static thread_local std::string g_create_err;

static constexpr int kManySlots = 8, kManyMax = 256;     // sk_fused_pass_many_dev: descriptor slots in the ring, batches per launch

static int fail(sk_ctx *c, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf; else g_create_err = buf;
	return code;
}

#define SK_HIP(c, call)                                                                            \
	do {                                                                                           \
		hipError_t e_ = (call);                                                                    \
		if (e_ != hipSuccess) return fail((c), SK_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
	} while (0)

static int bind(sk_ctx *c)
{
	SK_HIP(c, hipSetDevice(c->device));
	return SK_OK;
}

static int ensure_ws(sk_ctx *c, size_t bytes)
{
	if (bytes <= c->ws_bytes) return SK_OK;
	if (c->ws) { SK_HIP(c, hipStreamSynchronize(c->stream)); SK_HIP(c, hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
	size_t want = bytes + (bytes >> 2);
	hipError_t e = hipMalloc((void **)&c->ws, want);
	if (e != hipSuccess) { (void)hipGetLastError(); want = bytes; e = hipMalloc((void **)&c->ws, want); }      // the tolerated failure must not stay as the sticky last error
	if (e != hipSuccess) { (void)hipGetLastError(); c->ws = nullptr; return fail(c, SK_ERR_NOMEM, "device workspace of %zu bytes: %s", want, hipGetErrorString(e)); }
	c->ws_bytes = want;
	return SK_OK;
}

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

namespace sk {
hipStream_t ctx_stream(sk_ctx *c) { return c->stream; }
hipStream_t ctx_stream2(sk_ctx *c) { return c->stream2; }
int ctx_n_cu(sk_ctx *c) { return c->n_cu; }
// a buffer of at least `bytes` that stays with the ctx (slot 0..7; device memory or page-locked host memory); its contents are not kept
// when it has to grow.  nullptr + an error code in *rc when the allocation fails (the slot is then empty).
void *ctx_keep(sk_ctx *c, int slot, size_t bytes, bool pinned, int *rc)
{
	*rc = SK_OK;
	auto &k = c->kept[slot];
	if (k.p && k.cap >= bytes && k.pinned == pinned) return k.p;
	if (k.p) { if (k.pinned) (void)hipHostFree(k.p); else (void)hipFree(k.p); k.p = nullptr; k.cap = 0; }
	void *p = nullptr;
	const hipError_t e = pinned ? hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) : hipMalloc(&p, bytes ? bytes : 1);
	if (e != hipSuccess) { (void)hipGetLastError(); *rc = fail(c, SK_ERR_NOMEM, "%zu bytes of %s memory: %s", bytes, pinned ? "page-locked" : "device", hipGetErrorString(e)); return nullptr; }
	k.p = p; k.cap = bytes; k.pinned = pinned;
	return p;
}
size_t ctx_kept_bytes(sk_ctx *c, int slot) { return c->kept[slot].p ? c->kept[slot].cap : 0; }
void *ctx_ext(sk_ctx *c) { return c->ext; }
void ctx_set_ext(sk_ctx *c, void *p, void (*free_fn)(void *)) { c->ext = p; c->ext_free = free_fn; }
int ctx_bind(sk_ctx *c) { return bind(c); }
int ctx_fail(sk_ctx *c, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	return fail(c, code, "%s", buf);
}
}  // namespace sk

// The host-pointer entry points cut a batch into chunks and run them as a two-deep software pipeline: chunk k uses
// stream (k & 1) and half (k & 1) of the workspace, so the H2D copies of chunk k+1 run under the kernel and the D2H
// copies of chunk k, and a half is reused only by the next chunk of the SAME stream (in order: no hazard).  Copies are
// true DMA when the caller's buffers are pinned (sk_malloc_pinned); pageable buffers work too, staged by the runtime.
struct ChunkPipe {
	sk_ctx *c;
	size_t half = 0;
	int k = 0;
	explicit ChunkPipe(sk_ctx *ctx) : c(ctx) {}
	int begin(size_t bytes_per_chunk, size_t shared_tail = 0)      // shared_tail: bytes after the two halves used by every chunk (accumulators)
	{
		half = up256(bytes_per_chunk);
		if (int r = ensure_ws(c, 2 * half + shared_tail)) return r;
		SK_HIP(c, hipEventRecord(c->ev_pipe, c->stream));          // the second lane starts after whatever the ctx stream holds
		SK_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_pipe, 0));
		return SK_OK;
	}
	hipStream_t st() const { return (k & 1) ? c->stream2 : c->stream; }
	uint8_t *ws() const { return c->ws + (size_t)(k & 1) * half; }
	uint8_t *tail() const { return c->ws + 2 * half; }
	void next() { k++; }
	int end()
	{
		SK_HIP(c, hipStreamSynchronize(c->stream2));
		SK_HIP(c, hipStreamSynchronize(c->stream));
		return SK_OK;
	}
};
static const size_t kPipeChunkBytes = 48u << 20;
// chunk sizes of the host entry points; SK_HOST_CHUNK_LOG2 (read per call) shrinks every one of them to 2^k bytes / records,
// so that tests walk many chunks, both lanes and both workspace halves with small inputs
static size_t pipe_chunk(size_t dflt)
{
	if (const char *e = getenv("SK_HOST_CHUNK_LOG2")) { const int lg = atoi(e); if (lg >= 6 && lg <= 30) return (size_t)1 << lg; }
	return dflt;
}
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

extern "C" {

int sk_version(void) { return 0x000100; }

int sk_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int sk_create(int device_id, sk_ctx **out)
{
	if (!out) return fail(nullptr, SK_ERR_INVALID, "sk_create: out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) return fail(nullptr, SK_ERR_NO_DEVICE, "no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
	if (device_id < 0 || device_id >= n) return fail(nullptr, SK_ERR_NO_DEVICE, "device %d out of range (0..%d)", device_id, n - 1);
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device_id);
	if (e != hipSuccess) return fail(nullptr, SK_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, SK_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device_id, prop.gcnArchName);
	sk_ctx *c = new sk_ctx();
	c->device = device_id;
	c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	e = hipSetDevice(device_id);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pipe, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreate(&c->ev0);
	if (e == hipSuccess) e = hipEventCreate(&c->ev1);
	if (e != hipSuccess) { int r = fail(nullptr, SK_ERR_HIP, "context setup: %s", hipGetErrorString(e)); delete c; return r; }
	*out = c;
	return SK_OK;
}

void sk_destroy(sk_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
	if (c->d_raw) (void)hipFree(c->d_raw);
	if (c->d_onehot) (void)hipFree(c->d_onehot);
	if (c->d_lut) (void)hipFree(c->d_lut);
	if (c->d_bs) (void)hipFree(c->d_bs);
	if (c->d_nbr) (void)hipFree(c->d_nbr);
	if (c->d_counts) (void)hipFree(c->d_counts);
	if (c->d_counts_wide) (void)hipFree(c->d_counts_wide);
	if (c->d_count_rep) (void)hipFree(c->d_count_rep);
	if (c->ws) (void)hipFree(c->ws);
	if (c->pin) (void)hipHostFree(c->pin);
	for (auto &k : c->kept) if (k.p) { if (k.pinned) (void)hipHostFree(k.p); else (void)hipFree(k.p); }
	if (c->ext && c->ext_free) c->ext_free(c->ext);
	if (c->many_pin) (void)hipHostFree(c->many_pin);
	if (c->many_dev) (void)hipFree(c->many_dev);
	for (hipEvent_t e : c->many_ev) if (e) (void)hipEventDestroy(e);
	if (c->census) sk::census_destroy(c->census);
	if (c->comm) (void)sk_comm_destroy(c);
	if (c->d_cnt) (void)hipFree(c->d_cnt);
	if (c->d_genome) (void)hipFree(c->d_genome);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->ev_pipe) (void)hipEventDestroy(c->ev_pipe);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

const char *sk_last_error(const sk_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int sk_sync(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	SK_HIP(c, hipStreamSynchronize(c->stream));
	return SK_OK;
}

void *sk_stream(sk_ctx *c) { return c ? (void *)c->stream : nullptr; }

int sk_malloc_device(sk_ctx *c, size_t bytes, void **out)
{
	if (!c || !out) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	*out = nullptr;
	hipError_t e = hipMalloc(out, bytes ? bytes : 1);
	if (e != hipSuccess) return fail(c, SK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
	return SK_OK;
}
int sk_free_device(sk_ctx *c, void *p)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (p) SK_HIP(c, hipFree(p));
	return SK_OK;
}
// Pinned memory belongs to the process, not to a ctx (hipHostMallocPortable: page-locked for every device).  These two write
// NOTHING to the ctx — not even an error message: the hosts' worker threads call them on a ctx that another thread is running a
// pass on, and the message buffer of a ctx has one writer at a time — they return the code and sk_last_error is unchanged.
// They do make the ctx's device the calling thread's current one for the call (hipSetDevice is thread-local and touches no ctx
// state): on a thread that has no current device yet, the runtime would otherwise bring up a context on device 0 for the
// allocation.  The thread's own current device is put back before they return — a worker thread that also drives another ctx,
// another device or torch keeps launching where it was.
namespace {
struct ThreadDevice {              // the calling thread's current device for the lifetime of the object
	int saved = -1;
	bool ok;
	explicit ThreadDevice(int device)
	{
		if (hipGetDevice(&saved) != hipSuccess) { (void)hipGetLastError(); saved = -1; }
		ok = hipSetDevice(device) == hipSuccess;
		if (!ok) (void)hipGetLastError();
	}
	~ThreadDevice()
	{
		if (saved >= 0 && hipSetDevice(saved) != hipSuccess) (void)hipGetLastError();
	}
};
}  // namespace
int sk_malloc_pinned(sk_ctx *c, size_t bytes, void **out)
{
	if (!c || !out) return SK_ERR_INVALID;
	*out = nullptr;
	ThreadDevice td(c->device);
	if (!td.ok) return SK_ERR_HIP;
	if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return SK_ERR_NOMEM; }
	return SK_OK;
}
int sk_free_pinned(sk_ctx *c, void *p)
{
	if (!c) return SK_ERR_INVALID;
	if (!p) return SK_OK;
	ThreadDevice td(c->device);
	// (the pointer is freed even when the device could not be bound: portable pinned memory belongs to the process)
	if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return SK_ERR_HIP; }
	return td.ok ? SK_OK : SK_ERR_HIP;
}
int sk_copy_h2d(sk_ctx *c, void *dst, const void *src, size_t bytes)
{
	if (!c || (bytes && (!dst || !src))) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (bytes) SK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
	return SK_OK;
}
int sk_copy_d2h(sk_ctx *c, void *dst, const void *src, size_t bytes)
{
	if (!c || (bytes && (!dst || !src))) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (bytes) SK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
	return SK_OK;
}

// ---- barcode table ---------------------------------------------------------------------------------
int sk_set_barcodes(sk_ctx *c, const uint8_t *table, int S, int L, int max_diff)
{
	if (!c) return SK_ERR_INVALID;
	if (S < 0 || S > SK_MAX_SAMPLES) return fail(c, SK_ERR_INVALID, "S = %d outside 0..%d", S, SK_MAX_SAMPLES);
	if (L < 0 || L > SK_MAX_BARCODE_LEN || (S > 0 && L == 0)) return fail(c, SK_ERR_INVALID, "L = %d outside 1..%d", L, SK_MAX_BARCODE_LEN);
	if (S > 0 && !table) return fail(c, SK_ERR_INVALID, "table is NULL");
	if (max_diff < 0 || max_diff > 255) return fail(c, SK_ERR_INVALID, "max_diff = %d outside 0..255", max_diff);
	if (int r = bind(c)) return r;
	SK_HIP(c, hipStreamSynchronize(c->stream));
	if (c->d_raw) { SK_HIP(c, hipFree(c->d_raw)); c->d_raw = nullptr; }
	if (c->d_onehot) { SK_HIP(c, hipFree(c->d_onehot)); c->d_onehot = nullptr; }
	if (c->d_lut) { SK_HIP(c, hipFree(c->d_lut)); c->d_lut = nullptr; }
	if (c->d_bs) { SK_HIP(c, hipFree(c->d_bs)); c->d_bs = nullptr; }
	if (c->d_nbr) { SK_HIP(c, hipFree(c->d_nbr)); c->d_nbr = nullptr; }
	c->many_no_tab = nullptr;
	c->nbr = sk::LutDev{};
	c->bs_bytes = c->bs_mm_off = c->G = 0;
	if (c->d_counts) { SK_HIP(c, hipFree(c->d_counts)); c->d_counts = nullptr; }
	if (c->d_counts_wide) { SK_HIP(c, hipFree(c->d_counts_wide)); c->d_counts_wide = nullptr; }
	c->wide_dirty = false;
	if (c->d_count_rep) { SK_HIP(c, hipFree(c->d_count_rep)); c->d_count_rep = nullptr; }
	c->have_table = false;
	c->S = S; c->L = L; c->max_diff = max_diff;
	c->W = (L + 3) / 4;

	SK_HIP(c, hipMalloc((void **)&c->d_raw, (size_t)S * L + 16));
	if (S > 0) SK_HIP(c, hipMemcpy(c->d_raw, table, (size_t)S * L, hipMemcpyHostToDevice));

	// one-hot re-coding: possible when every position uses <= 7 distinct non-wildcard bytes
	bool onehot_ok = S > 0 && L <= sk::kMaxOneHotLen && !getenv("SK_NO_ONEHOT");
	std::vector<uint8_t> lut((size_t)(L ? L : 1) * 256, 0x80);
	std::vector<uint32_t> codes((size_t)(S ? S : 1) * (c->W ? c->W : 1), 0u);
	for (int k = 0; k < L && onehot_ok; k++) {
		int cls[256];
		for (int b = 0; b < 256; b++) cls[b] = -1;
		int ncls = 0;
		for (int s = 0; s < S; s++) {
			uint8_t b = table[(size_t)s * L + k];
			if (b == 'N' || b == 'U') continue;                 // wildcards: src/fasta_demultiplex.rs:273
			if (cls[b] < 0) { if (ncls == 7) { onehot_ok = false; break; } cls[b] = ncls++; }
		}
		if (!onehot_ok) break;
		for (int b = 0; b < 256; b++) lut[(size_t)k * 256 + b] = (uint8_t)(0x80 | (cls[b] >= 0 ? (1 << cls[b]) : 0));
		for (int s = 0; s < S; s++) {
			uint8_t b = table[(size_t)s * L + k];
			uint32_t code = (b == 'N' || b == 'U') ? 0x80u : (1u << cls[b]);
			codes[(size_t)s * c->W + (k >> 2)] |= code << (8 * (k & 3));
		}
	}
	if (onehot_ok) {
		SK_HIP(c, hipMalloc((void **)&c->d_lut, lut.size()));
		SK_HIP(c, hipMemcpy(c->d_lut, lut.data(), lut.size(), hipMemcpyHostToDevice));
		SK_HIP(c, hipMalloc((void **)&c->d_onehot, codes.size() * 4 + 64));
		SK_HIP(c, hipMemcpy(c->d_onehot, codes.data(), codes.size() * 4, hipMemcpyHostToDevice));
	}
	// bit-sliced matcher: possible when the whole sheet uses <= 7 distinct non-wildcard bytes and L <= 31
	{
		int cls[256];
		for (int b = 0; b < 256; b++) cls[b] = -1;
		int ncls = 0;
		bool ok = S > 0 && L <= sk::kMaxBitSlicedLen && !getenv("SK_NO_BITSLICE");
		for (int s = 0; s < S && ok; s++)
			for (int k = 0; k < L; k++) {
				uint8_t b = table[(size_t)s * L + k];
				if (b == 'N' || b == 'U') continue;
				if (cls[b] < 0) { if (ncls == 7) { ok = false; break; } cls[b] = ncls++; }
			}
		const int G = (S + 31) / 32;
		const int mm_off = 256 + ((G * 4 + 15) & ~15);
		const size_t bytes = (size_t)mm_off + (size_t)L * 8 * G * 4;
		if (ok && bytes <= (size_t)sk::kMaxBitSlicedBytes) {
			std::vector<uint8_t> blob(bytes, 0);
			for (int b = 0; b < 256; b++) blob[b] = (uint8_t)(cls[b] >= 0 ? cls[b] : 7);
			uint32_t *valid = (uint32_t *)(blob.data() + 256);
			uint32_t *mm = (uint32_t *)(blob.data() + mm_off);
			for (int s = 0; s < S; s++) {
				valid[s >> 5] |= 1u << (s & 31);
				for (int k = 0; k < L; k++) {
					uint8_t b = table[(size_t)s * L + k];
					if (b == 'N' || b == 'U') continue;                  // wildcard: never a mismatch
					for (int cc = 0; cc < 8; cc++)
						if (cc != cls[b]) mm[((size_t)k * 8 + cc) * G + (s >> 5)] |= 1u << (s & 31);
				}
			}
			SK_HIP(c, hipMalloc((void **)&c->d_bs, bytes));
			SK_HIP(c, hipMemcpy(c->d_bs, blob.data(), bytes, hipMemcpyHostToDevice));
			c->bs_bytes = (int)bytes; c->bs_mm_off = mm_off; c->G = G;
		}
	}
	// (the neighbourhood table of the sheet is built on the first demultiplex-alone call it can serve: ensure_neighbour_table)
	c->sheet.assign(table, table + (size_t)S * L);
	c->nbr_tried = false;
	SK_HIP(c, hipMalloc((void **)&c->d_counts, (size_t)(S + 3) * 8));
	SK_HIP(c, hipMemset(c->d_counts, 0, (size_t)(S + 3) * 8));
	if (S <= sk::kLutMaxSamples) {
		const size_t wide_bytes = S + 3 > sk::kCountDenseFrom ? (size_t)sk::kCountDenseRows * (S + 3) * 8
		                                                        : (((size_t)sk::kCountReplicas * (S + 3)) << sk::kCountWideShift) * 8;
		SK_HIP(c, hipMalloc((void **)&c->d_counts_wide, wide_bytes));
		SK_HIP(c, hipMemset(c->d_counts_wide, 0, wide_bytes));
	}
	if (S + 3 <= sk::kMaxLdsHist) {
		// two sets: the chunk pipeline of the host entry points has launches of its two lanes in flight together, and a
		// set serves one launch (the kernel and the fold behind it) at a time
		c->count_rep_pitch = (S + 3 + 31) & ~31;                 // whole 256-byte pieces: no line is shared by two copies
		c->count_rep_set = (size_t)sk::kCountReplicas * c->count_rep_pitch;
		SK_HIP(c, hipMalloc((void **)&c->d_count_rep, 2 * c->count_rep_set * 8));
		SK_HIP(c, hipMemset(c->d_count_rep, 0, 2 * c->count_rep_set * 8));
	}
	SK_HIP(c, hipDeviceSynchronize());     // the ctx stream is non-blocking: make the uploads visible to it
	c->have_table = true;
	return SK_OK;
}

// The neighbourhood table costs some host work per sheet (enumerate, decide, place): it is built when a call first
// asks for the decision alone or for the detail columns of matched rows only — a caller that always wants every row's
// detail (SK_DETAIL_FULL) never pays for it.  The caller has bound the ctx's device.
static int ensure_neighbour_table(sk_ctx *c)
{
	if (c->nbr_tried || !c->have_table) return SK_OK;
	c->nbr_tried = true;
	sk::LutHost h;
	// (SK_DEMUX_PAIR=0: never the factored form — a table too large for the LDS is then probed through the vector cache; tests)
	const char *env_pair = getenv("SK_DEMUX_PAIR");
	const int lds_budget = (env_pair && atoi(env_pair) == 0) ? 0 : (128 << 10);
	if (getenv("SK_NO_HASH_DEMUX") || !sk::lut_build(c->sheet.data(), c->S, c->L, c->max_diff, h, lds_budget)) return SK_OK;
	const size_t slot_bytes = h.slots.size() * 4, amb_bytes = h.amb.size() * 2;
	SK_HIP(c, hipMalloc((void **)&c->d_nbr, slot_bytes + amb_bytes + 16));
	SK_HIP(c, hipMemcpy(c->d_nbr, h.slots.data(), slot_bytes, hipMemcpyHostToDevice));
	if (amb_bytes) SK_HIP(c, hipMemcpy(c->d_nbr + slot_bytes, h.amb.data(), amb_bytes, hipMemcpyHostToDevice));
	SK_HIP(c, hipDeviceSynchronize());     // the ctx streams are non-blocking: make the upload visible to them
	c->nbr = h.dev;
	c->nbr_keys = (int64_t)h.n_keys; c->nbr_bytes = (int64_t)(slot_bytes + amb_bytes);
	if (h.dev.pair.bytes != 0) c->nbr.pair.tab = reinterpret_cast<const uint32_t *>(c->d_nbr);      // the factored form: three tables in one blob
	else c->nbr.tab = reinterpret_cast<const uint32_t *>(c->d_nbr);
	c->nbr.amb = reinterpret_cast<const int16_t *>(c->d_nbr + slot_bytes);
	return SK_OK;
}

int sk_barcode_table_info(sk_ctx *c, int *kind, int64_t *keys, int64_t *bytes)
{
	if (!c) return SK_ERR_INVALID;
	if (!c->have_table) return fail(c, SK_ERR_STATE, "sk_set_barcodes has not been called");
	if (int r = bind(c)) return r;
	if (int r = ensure_neighbour_table(c)) return r;
	const bool full = c->nbr.tab != nullptr, pair = c->nbr.pair.tab != nullptr;
	if (kind) *kind = (full ? SK_TABLE_FULL_KEY : pair ? SK_TABLE_FACTORED : SK_TABLE_NONE) | ((full || pair) && c->nbr.wide ? SK_TABLE_WIDE_CLASSES : 0);
	if (keys) *keys = (full || pair) ? c->nbr_keys : 0;
	if (bytes) *bytes = (full || pair) ? c->nbr_bytes : 0;
	return SK_OK;
}

static sk::BarcodeDev table_of(const sk_ctx *c)
{
	sk::BarcodeDev t;
	t.raw = c->d_raw; t.onehot = c->d_onehot; t.lut = c->d_lut;
	t.bs = c->d_bs; t.bs_bytes = c->bs_bytes; t.bs_mm_off = c->bs_mm_off; t.G = c->G;
	t.S = c->S; t.L = c->L; t.W = c->W; t.max_diff = c->max_diff;
	t.nbr = c->nbr;
	t.count_rep = c->d_count_rep; t.count_rep_pitch = c->count_rep_pitch;
	return t;
}

// The lookup kernel adds the ctx's counters into one line each (d_counts_wide); everything that reads or hands out
// d_counts first moves them over, on the ctx stream (the device is bound, and the second lane has been waited for).
static int fold_counts(sk_ctx *c)
{
	if (!c->wide_dirty) return SK_OK;
	SK_HIP(c, sk::launch_counts_fold_wide(c->d_counts_wide, c->S + 3, c->S + 3 > sk::kCountDenseFrom ? sk::kCountDenseRows : 0, c->d_counts, c->stream));
	c->wide_dirty = false;
	return SK_OK;
}

int sk_set_detail_mode(sk_ctx *c, int mode)
{
	if (!c) return SK_ERR_INVALID;
	if (mode != SK_DETAIL_FULL && mode != SK_DETAIL_MATCHED) return fail(c, SK_ERR_INVALID, "detail mode %d is neither SK_DETAIL_FULL nor SK_DETAIL_MATCHED", mode);
	c->detail_mode = mode;
	return SK_OK;
}

int sk_counts_reset(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (!c->have_table) return fail(c, SK_ERR_STATE, "sk_set_barcodes has not been called");
	if (int r = bind(c)) return r;
	if (int r = fold_counts(c)) return r;
	SK_HIP(c, hipMemsetAsync(c->d_counts, 0, (size_t)(c->S + 3) * 8, c->stream));
	return SK_OK;
}

int sk_counts_get(sk_ctx *c, uint64_t *counts)
{
	if (!c || !counts) return SK_ERR_INVALID;
	if (!c->have_table) return fail(c, SK_ERR_STATE, "sk_set_barcodes has not been called");
	if (int r = bind(c)) return r;
	if (int r = fold_counts(c)) return r;
	SK_HIP(c, hipMemcpyAsync(counts, c->d_counts, (size_t)(c->S + 3) * 8, hipMemcpyDeviceToHost, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));
	return SK_OK;
}

void *sk_counts_device_ptr(sk_ctx *c)
{
	if (!c || !c->have_table) return nullptr;
	if (bind(c) != SK_OK || fold_counts(c) != SK_OK) return nullptr;      // what was enqueued before this call is in the vector once the stream gets there
	return (void *)c->d_counts;
}

// ---- (e) the count reduce over RCCL ---------------------------------------------------------------------------
// librccl is 570 MB: the command-line hosts must not pay for loading it unless they drive several GPUs, so it is
// dlopen'ed on first use (the same SONAME torch loads, so a process that has torch in it shares its copy).
namespace {
struct Rccl {
	void *h = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string err;
};

Rccl *rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = {getenv("SK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *nm : names) {
			if (!nm || !*nm) continue;
			r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
			if (r.h) break;
			r.err = dlerror();
		}
		if (!r.h) return;
		bool ok = true;
		auto sym = [&](const char *nm) { void *p = dlsym(r.h, nm); if (!p) { ok = false; r.err = std::string("missing symbol ") + nm; } return p; };
		r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
		r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
		r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
		r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
		r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
		r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
		r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
		if (!ok) { dlclose(r.h); r.h = nullptr; }
	});
	return &r;
}

// communicators of one process over a set of devices (ncclCommInitAll), made once per device list
struct LocalComms { std::vector<int> devs; std::vector<ncclComm_t> comms; };
std::mutex g_local_m;
std::vector<LocalComms> g_local;

__global__ void counts_add_kernel(unsigned long long *dst, const unsigned long long *src, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] += src[i];
}
}  // namespace

// RCCL announces itself on stdout when a communicator comes up (version banner).  stdout belongs to the host's results
// (the reference's commands print there), so while a communicator is being made fd 1 points at stderr, and what stdio
// buffered during that time is flushed there before fd 1 is put back.
// fd 1 is process-wide state: one redirection at a time (two callers could otherwise put each other's descriptor back),
// and only the calls that make a communicator take it — a host thread that prints a result at that very moment would see
// it on stderr, which is why the command-line hosts make their communicators before their workers start.
// The lock is held only around the descriptor swaps, never across the RCCL call: two ranks of ONE process joining from two
// threads both block in the bootstrap until the other has arrived — the first redirection stays in place (a count) until the
// last of them is back.
static std::mutex g_stdout_m;
static int g_stdout_depth = 0, g_stdout_saved = -1;
struct StdoutToStderr {
	StdoutToStderr()
	{
		std::lock_guard<std::mutex> lk(g_stdout_m);
		if (g_stdout_depth++ == 0) { fflush(stdout); g_stdout_saved = dup(1); if (g_stdout_saved >= 0) (void)dup2(2, 1); }
	}
	~StdoutToStderr()
	{
		std::lock_guard<std::mutex> lk(g_stdout_m);
		if (--g_stdout_depth == 0) { fflush(stdout); if (g_stdout_saved >= 0) { (void)dup2(g_stdout_saved, 1); close(g_stdout_saved); g_stdout_saved = -1; } }
	}
};

#define SK_NCCL(c, call)                                                                                  \
	do {                                                                                                  \
		ncclResult_t r_ = (call);                                                                         \
		if (r_ != ncclSuccess) return fail((c), SK_ERR_COMM, "%s: %s", #call, rccl()->GetErrorString(r_)); \
	} while (0)

static int rccl_ready(sk_ctx *c)
{
	Rccl *r = rccl();
	if (!r->h) return fail(c, SK_ERR_COMM, "librccl.so.1 cannot be loaded (%s): the count reduce across GPUs needs RCCL", r->err.c_str());
	return SK_OK;
}

// A rank-local check that needs no other rank: can this ctx take part in an RCCL communicator at all (library loadable,
// device bindable)?  Hosts gather the answers of all ranks BEFORE any of them enters the blocking bootstrap.
int sk_comm_ready(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = rccl_ready(c)) return r;
	return bind(c);
}

int sk_comm_get_unique_id(uint8_t id[SK_COMM_ID_BYTES])
{
	static_assert(SK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
	if (!id) return SK_ERR_INVALID;
	if (int r = rccl_ready(nullptr)) return r;
	ncclUniqueId u;
	StdoutToStderr quiet;
	SK_NCCL(nullptr, rccl()->GetUniqueId(&u));
	memcpy(id, u.internal, SK_COMM_ID_BYTES);
	return SK_OK;
}

int sk_comm_init_rank(sk_ctx *c, const uint8_t id[SK_COMM_ID_BYTES], int rank, int n_ranks)
{
	if (!c || !id) return SK_ERR_INVALID;
	if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, SK_ERR_INVALID, "rank %d of %d", rank, n_ranks);
	if (c->comm) return fail(c, SK_ERR_STATE, "this ctx already has a communicator");
	if (int r = rccl_ready(c)) return r;
	if (int r = bind(c)) return r;
	ncclUniqueId u;
	memcpy(u.internal, id, SK_COMM_ID_BYTES);
	StdoutToStderr quiet;
	SK_NCCL(c, rccl()->CommInitRank(&c->comm, n_ranks, u, rank));
	c->comm_ranks = n_ranks;
	return SK_OK;
}

int sk_comm_destroy(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (!c->comm) return SK_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	ncclResult_t r = rccl()->CommDestroy(c->comm);
	c->comm = nullptr; c->comm_ranks = 0;
	return r == ncclSuccess ? SK_OK : fail(c, SK_ERR_COMM, "ncclCommDestroy: %s", rccl()->GetErrorString(r));
}

int sk_allreduce_u64_dev(sk_ctx *c, uint64_t *buf, size_t count)
{
	if (!c || (count && !buf)) return SK_ERR_INVALID;
	if (!c->comm || c->comm_ranks <= 1 || count == 0) return SK_OK;          // a world of one
	if (int r = bind(c)) return r;
	SK_NCCL(c, rccl()->AllReduce(buf, buf, count, ncclUint64, ncclSum, c->comm, c->stream));
	return SK_OK;
}

int sk_counts_allreduce(sk_ctx **ctxs, int n_ctx)
{
	if (!ctxs || n_ctx < 1) return SK_ERR_INVALID;
	for (int i = 0; i < n_ctx; i++) if (!ctxs[i]) return SK_ERR_INVALID;
	sk_ctx *c0 = ctxs[0];
	for (int i = 0; i < n_ctx; i++) {
		if (!ctxs[i]->have_table) return fail(c0, SK_ERR_STATE, "ctx %d: sk_set_barcodes has not been called", i);
		if (ctxs[i]->S != c0->S) return fail(c0, SK_ERR_INVALID, "ctx %d has %d samples, ctx 0 has %d", i, ctxs[i]->S, c0->S);
	}
	const int nc = c0->S + 3;
	for (int i = 0; i < n_ctx; i++) {
		if (int r = bind(ctxs[i])) return r;
		if (int r = fold_counts(ctxs[i])) return r;
	}
	if (n_ctx == 1) return sk_allreduce_u64_dev(c0, (uint64_t *)c0->d_counts, (size_t)nc);     // across ranks (or nothing to do)
	for (int i = 0; i < n_ctx; i++)
		if (ctxs[i]->comm) return fail(c0, SK_ERR_STATE, "ctx %d belongs to a one-process-per-GPU communicator: reduce it alone", i);

	// 1. everything enqueued so far must have happened before another stream reads the counters
	for (int i = 0; i < n_ctx; i++) { SK_HIP(c0, hipSetDevice(ctxs[i]->device)); SK_HIP(c0, hipStreamSynchronize(ctxs[i]->stream)); }
	// 2. ctxs that share a device: summed on that device into the first of them (its "leader")
	std::vector<int> leader_of(n_ctx), leaders;
	for (int i = 0; i < n_ctx; i++) {
		int l = -1;
		for (int k : leaders) if (ctxs[k]->device == ctxs[i]->device) { l = k; break; }
		if (l < 0) { leaders.push_back(i); l = i; }
		leader_of[i] = l;
	}
	for (int i = 0; i < n_ctx; i++) {
		const int l = leader_of[i];
		if (l == i) continue;
		SK_HIP(c0, hipSetDevice(ctxs[l]->device));
		counts_add_kernel<<<(nc + 255) / 256, 256, 0, ctxs[l]->stream>>>(ctxs[l]->d_counts, ctxs[i]->d_counts, nc);
		SK_HIP(c0, hipGetLastError());
	}
	// 3. distinct devices: one RCCL all-reduce (sum, u64) on the leaders' streams
	if (leaders.size() > 1) {
		if (int r = rccl_ready(c0)) return r;
		std::vector<int> devs;
		for (int k : leaders) devs.push_back(ctxs[k]->device);
		std::vector<ncclComm_t> comms;
		// the communicators of a device set are shared by every caller that reduces over it: the lock is held from their
		// lookup to the end of the grouped call, so two threads never interleave their operations on one ncclComm
		std::lock_guard<std::mutex> lk(g_local_m);
		{
			for (const LocalComms &lc : g_local) if (lc.devs == devs) { comms = lc.comms; break; }
			if (comms.empty()) {
				comms.resize(devs.size());
				StdoutToStderr quiet;
				SK_NCCL(c0, rccl()->CommInitAll(comms.data(), (int)devs.size(), devs.data()));
				g_local.push_back({devs, comms});
			}
		}
		SK_NCCL(c0, rccl()->GroupStart());
		for (size_t k = 0; k < leaders.size(); k++) {
			sk_ctx *l = ctxs[leaders[k]];
			(void)hipSetDevice(l->device);                         // each rank's call with its own device current
			ncclResult_t r = rccl()->AllReduce(l->d_counts, l->d_counts, (size_t)nc, ncclUint64, ncclSum, comms[k], l->stream);
			if (r != ncclSuccess) { (void)rccl()->GroupEnd(); return fail(c0, SK_ERR_COMM, "ncclAllReduce: %s", rccl()->GetErrorString(r)); }
		}
		SK_NCCL(c0, rccl()->GroupEnd());
	}
	// 4. the totals back to the ctxs that share a leader's device
	for (int i = 0; i < n_ctx; i++) {
		const int l = leader_of[i];
		if (l == i) continue;
		SK_HIP(c0, hipSetDevice(ctxs[l]->device));
		SK_HIP(c0, hipMemcpyAsync(ctxs[i]->d_counts, ctxs[l]->d_counts, (size_t)nc * 8, hipMemcpyDeviceToDevice, ctxs[l]->stream));
	}
	for (int k : leaders) { SK_HIP(c0, hipSetDevice(ctxs[k]->device)); SK_HIP(c0, hipStreamSynchronize(ctxs[k]->stream)); }
	return SK_OK;
}

// ---- fused pass ------------------------------------------------------------------------------------
static int check_fused(sk_ctx *c, const sk_fused_args *a, bool dev)
{
	if (!a) return fail(c, SK_ERR_INVALID, "args is NULL");
	if (a->n < 0) return fail(c, SK_ERR_INVALID, "n = %lld is negative", (long long)a->n);
	if (a->n_mates < 0 || a->n_mates > 2) return fail(c, SK_ERR_INVALID, "n_mates = %d outside 0..2", a->n_mates);
	bool any = false;
	for (int m = 0; m < a->n_mates; m++) {
		const sk_mate &mt = a->mate[m];
		if (!mt.out_seq && !mt.lowest_k) continue;
		any = true;
		if (a->stride <= 0 || a->stride > 65535) return fail(c, SK_ERR_INVALID, "stride = %d outside 1..65535", a->stride);
		if (!mt.qual) return fail(c, SK_ERR_INVALID, "mate %d: qual is NULL", m);
		if (mt.out_seq && !mt.seq) return fail(c, SK_ERR_INVALID, "mate %d: out_seq given without seq", m);
		if (dev && (!aligned16(mt.qual) || (mt.out_seq && (!aligned16(mt.seq) || !aligned16(mt.out_seq)))))
			return fail(c, SK_ERR_INVALID, "mate %d: device byte matrices must be 16-byte aligned", m);
	}
	if (a->bc) {
		if (!c->have_table) return fail(c, SK_ERR_STATE, "sk_set_barcodes has not been called");
		if (a->bc_stride < c->L || a->bc_stride <= 0) return fail(c, SK_ERR_INVALID, "bc_stride = %d is below the barcode length %d", a->bc_stride, c->L);
		if (a->bc_stride > sk::kMaxTileStride) return fail(c, SK_ERR_INVALID, "bc_stride = %d above %d", a->bc_stride, sk::kMaxTileStride);
		if (!a->assign) return fail(c, SK_ERR_INVALID, "assign is NULL");
		if (dev && !aligned16(a->bc)) return fail(c, SK_ERR_INVALID, "bc must be 16-byte aligned");
		// the lookup kernels write a wave's 256 rows of every output column with ONE wide store per lane (16 bytes of assign, 4 of
		// lowest_diff, 8 of first_idx / last_idx): the columns' natural alignment is not enough for a device pointer
		if (dev && (!aligned16(a->assign) || ((uintptr_t)a->lowest_diff & 3u) || ((uintptr_t)a->first_idx & 7u) || ((uintptr_t)a->last_idx & 7u)))
			return fail(c, SK_ERR_INVALID, "device output columns: assign must be 16-byte, lowest_diff 4-byte, first_idx / last_idx 8-byte aligned");
	} else if (!any) {
		return fail(c, SK_ERR_INVALID, "nothing to do: no bc, no out_seq, no lowest_k");
	}
	return SK_OK;
}

// demultiplex alone, with the decision only or the detail columns of matched rows only: the neighbourhood table's case.
// Called with the ctx's device bound (the table is uploaded to the current device).
static int prepare_demux(sk_ctx *c, const sk_fused_args *a)
{
	if (!a->bc) return SK_OK;
	// (a pass with mates matches in the tile pass itself when the bit-sliced matcher applies — at most 128 samples, the tile's
	// barcodes in two register chunks: launch_tile_pass's fuse_demux; otherwise its barcode phase is a launch of its own and
	// takes the table like a demultiplex-alone call)
	bool any_mate = false;
	for (int m = 0; m < a->n_mates; m++) any_mate = any_mate || a->mate[m].out_seq || a->mate[m].lowest_k;
	if (sk::tile_pass_fuses_demux(true, any_mate, a->stride, c->d_bs != nullptr, c->G, c->S, a->bc_stride)) return SK_OK;
	const bool want_detail = a->lowest_diff || a->first_idx || a->last_idx;
	if (want_detail && c->detail_mode != SK_DETAIL_MATCHED) return SK_OK;
	return ensure_neighbour_table(c);
}

static sk::TileArgs tile_args_of(const sk_ctx *c, const sk_fused_args *a)
{
	sk::TileArgs t;
	memset(&t, 0, sizeof t);
	t.n = a->n; t.n_mates = a->n_mates; t.stride = a->stride;
	t.qc = sk::make_qual_consts(a->min_baseq);
	for (int m = 0; m < a->n_mates; m++) {
		t.mate[m].seq = a->mate[m].seq; t.mate[m].qual = a->mate[m].qual; t.mate[m].len = a->mate[m].len;
		t.mate[m].out_seq = a->mate[m].out_seq; t.mate[m].lowest_k = a->mate[m].lowest_k;
	}
	t.bc = a->bc; t.bc_stride = a->bc_stride;
	if (a->bc) t.table = table_of(c);
	t.assign = a->assign; t.lowest_diff = a->lowest_diff; t.first_idx = a->first_idx; t.last_idx = a->last_idx;
	t.detail_matched = c->detail_mode == SK_DETAIL_MATCHED ? 1 : 0;
	t.counts = a->counts ? (unsigned long long *)a->counts : c->d_counts;
	t.counts_wide = a->counts ? nullptr : c->d_counts_wide;
	t.counts_wide_rows = (t.counts_wide && c->S + 3 > sk::kCountDenseFrom) ? sk::kCountDenseRows : 0;
	return t;
}

int sk_fused_pass_dev(sk_ctx *c, const sk_fused_args *a)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;          // first: what follows may upload the sheet's table to the current device
	if (int r = check_fused(c, a, true)) return r;
	if (a->n == 0) return SK_OK;
	if (int r = prepare_demux(c, a)) return r;
	sk::TileArgs t = tile_args_of(c, a);
	if (t.bc && t.counts_wide) c->wide_dirty = true;
	SK_HIP(c, sk::launch_tile_pass(t, c->n_cu, c->stream));
	return SK_OK;
}

static void one_mate(sk_fused_args &a, const uint8_t *seq, const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                     uint8_t min_baseq, uint8_t *out_seq, uint16_t *lowest_k);

// Many independent batches in one call: every batch is checked, the sheet's table goes up once, and the launches
// follow each other on the ctx stream with nothing between them — what a host that has its batches at hand gets without a call,
// an argument check and an event per batch (bench.py: `frac_many` against `frac`, one event pair per call).
// Lookups of one shape (demultiplex alone, a table in LDS) run as ONE launch: the steps of all batches are dealt to the launch's
// waves, the table is staged once, the next batch's rows are fetched while the last of this one are looked up (ManyBatch,
// demux_lut8x2_kernel).  Measured before that (round 6, gpurun_out/r06_l): the batches dealt to four streams behind one fork event
// — SLOWER than one stream (10 M x 8 bp: 0.47 of the HBM peak against 0.58 per call and 0.62 back to back): two lookup kernels
// on a CU are thirty-two waves where sixteen stream best, and every batch pays two more events.
int sk_fused_pass_many_dev(sk_ctx *c, const sk_fused_args *batches, int n_batches)
{
	if (!c) return SK_ERR_INVALID;
	if (n_batches < 0 || (n_batches > 0 && !batches)) return fail(c, SK_ERR_INVALID, "n_batches = %d", n_batches);
	if (n_batches == 0) return SK_OK;
	if (int r = bind(c)) return r;
	for (int i = 0; i < n_batches; i++) if (int r = check_fused(c, &batches[i], true)) return r;
	// ONE launch for all of them when they are lookups of one shape: demultiplex alone into the ctx's counters, the same pitch and the
	// same set of output columns, served by a table in LDS (launch_tile_pass says no for any other shape: the loop below)
	{
		bool same = n_batches >= 2;
		int64_t rows = 0;
		for (int i = 0; i < n_batches && same; i++) {
			const sk_fused_args &a = batches[i], &f = batches[0];
			bool mates = false;
			for (int m = 0; m < a.n_mates; m++) mates = mates || a.mate[m].out_seq || a.mate[m].lowest_k;
			same = !mates && a.bc && !a.counts && a.bc_stride == f.bc_stride && !a.lowest_diff == !f.lowest_diff && !a.first_idx == !f.first_idx && !a.last_idx == !f.last_idx;
			rows += ((a.n + 255) / 256) * 256;
		}
		if (const char *ev = getenv("SK_MANY_ONE_LAUNCH")) same = same && atoi(ev) != 0;      // (A/B: the batches' own launches instead)
		const int want_detail_key = same ? ((batches[0].lowest_diff ? 1 : 0) | (batches[0].first_idx ? 2 : 0) | (batches[0].last_idx ? 4 : 0) | (c->detail_mode << 3)) : 0;
		if (same && c->many_no_tab && c->many_no_tab == (const void *)c->d_nbr && c->many_no_stride == batches[0].bc_stride && c->many_no_detail == want_detail_key && !getenv("SK_LUT_MANY_GATHER")) same = false;
		if (same && rows > 0 && rows < ((int64_t)1 << 31) && n_batches <= kManyMax) {
			if (int r = prepare_demux(c, &batches[0])) return r;
			if (!c->many_pin) {
				SK_HIP(c, hipHostMalloc((void **)&c->many_pin, kManySlots * kManyMax * sizeof(sk::ManyBatch), hipHostMallocDefault));
				SK_HIP(c, hipMalloc((void **)&c->many_dev, kManySlots * kManyMax * sizeof(sk::ManyBatch)));
				for (int k = 0; k < kManySlots; k++) SK_HIP(c, hipEventCreateWithFlags(&c->many_ev[k], hipEventDisableTiming));
			}
			const int slot = c->many_next++ % kManySlots;
			SK_HIP(c, hipEventSynchronize(c->many_ev[slot]));                // (the launch that read this slot eight calls ago: long done)
			sk::ManyBatch *hb = c->many_pin + (size_t)slot * kManyMax, *db = c->many_dev + (size_t)slot * kManyMax;
			int64_t q0 = 0;
			int nb = 0;
			for (int i = 0; i < n_batches; i++) {
				const sk_fused_args &a = batches[i];
				if (a.n == 0) continue;
				const int64_t r0 = q0 * 256;
				sk::ManyBatch &m = hb[nb++];
				m.bc = a.bc - r0 * a.bc_stride;
				m.assign = a.assign - r0;
				m.lowest_diff = a.lowest_diff ? a.lowest_diff - r0 : nullptr;
				m.first_idx = a.first_idx ? a.first_idx - r0 : nullptr;
				m.last_idx = a.last_idx ? a.last_idx - r0 : nullptr;
				m.row_end = (uint32_t)(r0 + a.n);
				q0 += (a.n + 255) / 256;
				m.q_end = (int32_t)q0;
			}
			if (nb > 0) {
				// (the kernel reads the descriptors where they lie, in pinned host memory: a wave reads one when its cursor moves to the next
				// batch — a copy of 200 bytes in front of the launch was 10 us on the stream, a tenth of what four batches take)
				(void)db;
				sk_fused_args first = batches[0];
				for (int i = 0; i < n_batches; i++) if (batches[i].n > 0) { first = batches[i]; break; }
				sk::TileArgs t = tile_args_of(c, &first);
				t.n = q0 * 256;
				t.many = hb; t.n_many = nb; t.many_quads = (int)q0;
				if (t.counts_wide) c->wide_dirty = true;
				const hipError_t e = sk::launch_tile_pass(t, c->n_cu, c->stream);
				SK_HIP(c, hipEventRecord(c->many_ev[slot], c->stream));
				if (e == hipSuccess) return SK_OK;
				if (e != hipErrorNotSupported) return fail(c, SK_ERR_HIP, "launch_tile_pass (many batches): %s", hipGetErrorString(e));
				(void)hipGetLastError();
				if (t.n >= 1500000) { c->many_no_tab = (const void *)c->d_nbr; c->many_no_stride = batches[0].bc_stride; c->many_no_detail = want_detail_key; }      // (below that the answer may depend on the size)
			} else return SK_OK;
		}
	}
	// batches without a barcode phase (trim / mask alone: nothing but their own outputs is written, no counters) alternate between the
	// ctx's two streams, so that one batch's last tiles and the next one's first share the chip: cfg 2's 1 M-read batches from HBM 39.0 ->
	// 31.5 us each = 0.49 -> 0.60 of the HBM peak (SK_MANY_TWO_STREAMS=0: one stream).  (Batches with a barcode phase stay on one stream:
	// their launches fold per-workgroup counter copies that the next launch is already adding to.)
	bool two = n_batches >= 2;
	int two_mode = 1;
	if (const char *ev = getenv("SK_MANY_TWO_STREAMS")) { two_mode = atoi(ev); two = two && two_mode != 0; }
	for (int i = 0; i < n_batches && two; i++) {
		if (batches[i].bc == nullptr) continue;
		// (A/B, SK_MANY_TWO_STREAMS=2: barcode-only batches whose launches take the lookup table and add to the ctx's wide counters)
		bool mates = false;
		for (int m = 0; m < batches[i].n_mates; m++) mates = mates || batches[i].mate[m].out_seq || batches[i].mate[m].lowest_k;
		if (two_mode < 2 || mates || batches[i].counts) { two = false; break; }
		if (int r = prepare_demux(c, &batches[i])) return r;
		const sk::TileArgs t = tile_args_of(c, &batches[i]);
		two = t.counts_wide != nullptr && sk::tile_pass_demux_by_table(t);
	}
	if (two) {
		SK_HIP(c, hipEventRecord(c->ev_pipe, c->stream));
		SK_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_pipe, 0));
	}
	int k = 0;
	for (int i = 0; i < n_batches; i++) {
		const sk_fused_args *a = &batches[i];
		if (a->n == 0) continue;
		if (int r = prepare_demux(c, a)) return r;
		sk::TileArgs t = tile_args_of(c, a);
		if (t.bc && t.counts_wide) c->wide_dirty = true;
		SK_HIP(c, sk::launch_tile_pass(t, c->n_cu, (two && (k & 1)) ? c->stream2 : c->stream));
		k++;
	}
	if (two) {
		SK_HIP(c, hipEventRecord(c->ev_pipe, c->stream2));
		SK_HIP(c, hipStreamWaitEvent(c->stream, c->ev_pipe, 0));
	}
	return SK_OK;
}

int sk_demux_assign_many_dev(sk_ctx *c, const sk_demux_batch *b, int n_batches, int bc_stride)
{
	if (!c) return SK_ERR_INVALID;
	if (n_batches < 0 || (n_batches > 0 && !b)) return fail(c, SK_ERR_INVALID, "n_batches = %d", n_batches);
	std::vector<sk_fused_args> v((size_t)n_batches);
	for (int i = 0; i < n_batches; i++) {
		sk_fused_args &a = v[(size_t)i];
		memset(&a, 0, sizeof a);
		if (!b[i].bc && b[i].n > 0) return fail(c, SK_ERR_INVALID, "batch %d: bc is NULL", i);
		a.n = b[i].n; a.bc = b[i].bc; a.bc_stride = bc_stride; a.assign = b[i].assign;
		a.lowest_diff = b[i].lowest_diff; a.first_idx = b[i].first_idx; a.last_idx = b[i].last_idx;
	}
	return sk_fused_pass_many_dev(c, v.data(), n_batches);
}

int sk_trim_by_quality_many_dev(sk_ctx *c, const sk_trim_batch *b, int n_batches, int stride, uint8_t min_baseq)
{
	if (!c) return SK_ERR_INVALID;
	if (n_batches < 0 || (n_batches > 0 && !b)) return fail(c, SK_ERR_INVALID, "n_batches = %d", n_batches);
	std::vector<sk_fused_args> v((size_t)n_batches);
	for (int i = 0; i < n_batches; i++) {
		if (b[i].n > 0 && !b[i].lowest_k) return fail(c, SK_ERR_INVALID, "batch %d: lowest_k is NULL", i);
		one_mate(v[(size_t)i], nullptr, b[i].qual, b[i].len, stride, b[i].n, min_baseq, nullptr, b[i].lowest_k);
	}
	return sk_fused_pass_many_dev(c, v.data(), n_batches);
}

// Host-pointer form: chunks of rows are staged through the device workspace.
int sk_fused_pass(sk_ctx *c, const sk_fused_args *a)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (int r = check_fused(c, a, false)) return r;
	if (a->n == 0) return SK_OK;
	if (int r = prepare_demux(c, a)) return r;
	const int64_t stride = a->stride;
	// bytes of workspace per row
	size_t per_row = 0;
	for (int m = 0; m < a->n_mates; m++) {
		const sk_mate &mt = a->mate[m];
		if (!mt.out_seq && !mt.lowest_k) continue;
		per_row += (size_t)stride;                         // qual
		if (mt.out_seq) per_row += 2 * (size_t)stride;     // seq + out
		if (mt.len) per_row += 2;
		if (mt.lowest_k) per_row += 2;
	}
	if (a->bc) per_row += (size_t)a->bc_stride + 4 + 1 + 2 + 2;
	int64_t chunk = (int64_t)(pipe_chunk(kPipeChunkBytes) / (per_row ? per_row : 1));
	chunk &= ~(int64_t)63;
	if (chunk < 64) chunk = 64;
	if (chunk > a->n) chunk = (a->n + 63) & ~(int64_t)63;
	ChunkPipe pipe(c);
	if (int r = pipe.begin((size_t)chunk * per_row + 64 * 256)) return r;

	for (int64_t r0 = 0; r0 < a->n; r0 += chunk, pipe.next()) {
		const int64_t nr = (a->n - r0) < chunk ? (a->n - r0) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *p = pipe.ws();
		auto carve = [&](size_t bytes) { uint8_t *q = p; p += up256(bytes); return q; };
		sk_fused_args d = *a;
		d.n = nr; d.counts = nullptr;
		for (int m = 0; m < a->n_mates; m++) {
			const sk_mate &mt = a->mate[m];
			sk_mate &dm = d.mate[m];
			dm = sk_mate{nullptr, nullptr, nullptr, nullptr, nullptr};
			if (!mt.out_seq && !mt.lowest_k) continue;
			uint8_t *dq = carve((size_t)nr * stride);
			SK_HIP(c, hipMemcpyAsync(dq, mt.qual + r0 * stride, (size_t)nr * stride, hipMemcpyHostToDevice, st));
			dm.qual = dq;
			if (mt.out_seq) {
				uint8_t *ds = carve((size_t)nr * stride);
				SK_HIP(c, hipMemcpyAsync(ds, mt.seq + r0 * stride, (size_t)nr * stride, hipMemcpyHostToDevice, st));
				dm.seq = ds;
				dm.out_seq = carve((size_t)nr * stride);
			}
			if (mt.len) {
				uint16_t *dl = (uint16_t *)carve((size_t)nr * 2);
				SK_HIP(c, hipMemcpyAsync(dl, mt.len + r0, (size_t)nr * 2, hipMemcpyHostToDevice, st));
				dm.len = dl;
			}
			if (mt.lowest_k) dm.lowest_k = (uint16_t *)carve((size_t)nr * 2);
		}
		if (a->bc) {
			uint8_t *db = carve((size_t)nr * a->bc_stride);
			SK_HIP(c, hipMemcpyAsync(db, a->bc + r0 * a->bc_stride, (size_t)nr * a->bc_stride, hipMemcpyHostToDevice, st));
			d.bc = db;
			d.assign = (int32_t *)carve((size_t)nr * 4);
			d.lowest_diff = a->lowest_diff ? carve((size_t)nr) : nullptr;
			d.first_idx = a->first_idx ? (int16_t *)carve((size_t)nr * 2) : nullptr;
			d.last_idx = a->last_idx ? (int16_t *)carve((size_t)nr * 2) : nullptr;
		}
		sk::TileArgs t = tile_args_of(c, &d);
		if (t.table.count_rep && (pipe.k & 1)) t.table.count_rep += c->count_rep_set;     // the second lane's set of counter copies
		if (t.bc && t.counts_wide) c->wide_dirty = true;
		SK_HIP(c, sk::launch_tile_pass(t, c->n_cu, st));
		for (int m = 0; m < a->n_mates; m++) {
			const sk_mate &mt = a->mate[m];
			const sk_mate &dm = d.mate[m];
			if (mt.out_seq) SK_HIP(c, hipMemcpyAsync(mt.out_seq + r0 * stride, dm.out_seq, (size_t)nr * stride, hipMemcpyDeviceToHost, st));
			if (mt.lowest_k) SK_HIP(c, hipMemcpyAsync(mt.lowest_k + r0, dm.lowest_k, (size_t)nr * 2, hipMemcpyDeviceToHost, st));
		}
		if (a->bc) {
			SK_HIP(c, hipMemcpyAsync(a->assign + r0, d.assign, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
			if (a->lowest_diff) SK_HIP(c, hipMemcpyAsync(a->lowest_diff + r0, d.lowest_diff, (size_t)nr, hipMemcpyDeviceToHost, st));
			if (a->first_idx) SK_HIP(c, hipMemcpyAsync(a->first_idx + r0, d.first_idx, (size_t)nr * 2, hipMemcpyDeviceToHost, st));
			if (a->last_idx) SK_HIP(c, hipMemcpyAsync(a->last_idx + r0, d.last_idx, (size_t)nr * 2, hipMemcpyDeviceToHost, st));
		}
	}
	return pipe.end();
}

// ---- placement tuning -------------------------------------------------------------------------------------------
int sk_fused_tune_placement_dev(sk_ctx *c, sk_fused_args *a, const sk_fused_candidates *cd, int sweeps, float *ms_before, float *ms_after, int *n_probes)
{
	if (!c) return SK_ERR_INVALID;
	if (!a || !cd) return fail(c, SK_ERR_INVALID, "args or candidates is NULL");
	if (cd->k < 1 || cd->k > SK_MAX_CANDIDATES || sweeps < 0) return fail(c, SK_ERR_INVALID, "k = %d (1..%d), sweeps = %d", cd->k, SK_MAX_CANDIDATES, sweeps);
	if (int r = bind(c)) return r;
	if (int r = check_fused(c, a, true)) return r;
	if (int r = prepare_demux(c, a)) return r;
	// the matrices that take part: those the pass uses and that have candidates
	struct Slot { int mate, what; };                       // what: 0 seq, 1 qual, 2 out_seq
	std::vector<Slot> slots;
	for (int m = 0; m < a->n_mates; m++) {
		const sk_mate &mt = a->mate[m];
		if (!mt.out_seq && !mt.lowest_k) continue;
		if (mt.out_seq && cd->seq[m][0]) slots.push_back({m, 0});
		if (cd->qual[m][0]) slots.push_back({m, 1});
		if (mt.out_seq && cd->out_seq[m][0]) slots.push_back({m, 2});
	}
	for (const Slot &sl : slots)
		for (int k = 0; k < cd->k; k++) {
			const void *p = sl.what == 0 ? (const void *)cd->seq[sl.mate][k] : sl.what == 1 ? (const void *)cd->qual[sl.mate][k] : (const void *)cd->out_seq[sl.mate][k];
			if (!p || !aligned16(p)) return fail(c, SK_ERR_INVALID, "candidate %d of mate %d is NULL or not 16-byte aligned", k, sl.mate);
		}
	unsigned long long *scratch = nullptr;                  // the probes count here, not into the caller's counters
	SK_HIP(c, hipMalloc((void **)&scratch, (size_t)(c->S + 3) * 8 + 8));
	sk_fused_args t = *a;
	t.counts = (uint64_t *)scratch;
	std::vector<int> choice(slots.size(), 0);
	auto apply = [&](sk_fused_args &x, const std::vector<int> &ch) {
		for (size_t i = 0; i < slots.size(); i++) {
			sk_mate &mt = x.mate[slots[i].mate];
			if (slots[i].what == 0) mt.seq = cd->seq[slots[i].mate][ch[i]];
			else if (slots[i].what == 1) mt.qual = cd->qual[slots[i].mate][ch[i]];
			else mt.out_seq = cd->out_seq[slots[i].mate][ch[i]];
		}
	};
	int probes = 0;
	int rc = SK_OK;
	auto time_choice = [&](const std::vector<int> &ch, float &ms) -> int {
		apply(t, ch);
		sk::TileArgs ta = tile_args_of(c, &t);
		probes++;
		hipError_t e = sk::launch_tile_pass(ta, c->n_cu, c->stream);                 // warm
		if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
		for (int i = 0; i < 2 && e == hipSuccess; i++) e = sk::launch_tile_pass(ta, c->n_cu, c->stream);
		if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
		if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
		if (e != hipSuccess) return fail(c, SK_ERR_HIP, "placement probe: %s", hipGetErrorString(e));
		ms *= 0.5f;
		return SK_OK;
	};
	float best = 0.f;
	rc = time_choice(choice, best);
	const float first = best;
	for (int sw = 0; sw < sweeps && rc == SK_OK; sw++) {
		bool moved = false;
		for (size_t i = 0; i < slots.size() && rc == SK_OK; i++)
			for (int k = 0; k < cd->k && rc == SK_OK; k++) {
				if (k == choice[i]) continue;
				std::vector<int> trial = choice;
				trial[i] = k;
				float ms = 0.f;
				rc = time_choice(trial, ms);
				if (rc == SK_OK && ms < best * 0.998f) { best = ms; choice = trial; moved = true; }
			}
		if (!moved) break;
	}
	(void)hipStreamSynchronize(c->stream);
	(void)hipFree(scratch);
	if (rc != SK_OK) return rc;
	apply(*a, choice);
	if (ms_before) *ms_before = first;
	if (ms_after) *ms_after = best;
	if (n_probes) *n_probes = probes;
	return SK_OK;
}

// ---- tile-blocked batches ---------------------------------------------------------------------------------------
int sk_blocked_layout_init(sk_blocked_layout *lay, int n_mates, int stride, int bc_stride, int flags)
{
	if (!lay) return SK_ERR_INVALID;
	if (n_mates < 1 || n_mates > 2 || bc_stride < 0 || stride <= 0 || stride > 65535) return SK_ERR_INVALID;
	if (!(flags & (SK_BLK_MASK | SK_BLK_TRIM))) return SK_ERR_INVALID;      // barcodes alone: sk_demux_assign(_dev)
	memset(lay, 0, sizeof *lay);
	lay->n_mates = n_mates; lay->stride = stride; lay->bc_stride = bc_stride; lay->flags = flags;
	for (int m = 0; m < 2; m++) lay->in_qual[m] = lay->in_seq[m] = lay->in_len[m] = lay->out_seq[m] = lay->out_lowest_k[m] = -1;
	lay->in_bc = lay->out_assign = lay->out_lowest_diff = lay->out_first_idx = lay->out_last_idx = -1;
	int64_t in = 0, out = 0;
	auto seg = [](int64_t &at, int64_t bytes) { int64_t o = at; at += (bytes + 63) & ~(int64_t)63; return (int32_t)o; };
	for (int m = 0; m < n_mates; m++) {                       // streams first, in the order the kernel walks them
		lay->in_qual[m] = seg(in, 64 * (int64_t)stride);
		if (flags & SK_BLK_MASK) lay->in_seq[m] = seg(in, 64 * (int64_t)stride);
	}
	if (bc_stride > 0) lay->in_bc = seg(in, 64 * (int64_t)bc_stride);
	if (flags & SK_BLK_LEN) for (int m = 0; m < n_mates; m++) lay->in_len[m] = seg(in, 128);
	if (flags & ~(SK_BLK_MASK | SK_BLK_TRIM | SK_BLK_LEN | SK_BLK_DETAIL)) return SK_ERR_INVALID;
	if (flags & SK_BLK_MASK) for (int m = 0; m < n_mates; m++) lay->out_seq[m] = seg(out, 64 * (int64_t)stride);
	if (flags & SK_BLK_TRIM) for (int m = 0; m < n_mates; m++) lay->out_lowest_k[m] = seg(out, 128);
	if (bc_stride > 0) {
		lay->out_assign = seg(out, 256);
		if (flags & SK_BLK_DETAIL) { lay->out_lowest_diff = seg(out, 64); lay->out_first_idx = seg(out, 128); lay->out_last_idx = seg(out, 128); }
	}
	in = (in + 127) & ~(int64_t)127;                            // whole 128-byte lines: no line is shared by two tiles (two waves would fetch it)
	out = (out + 127) & ~(int64_t)127;
	if (in > 0x3fffffff || out > 0x3fffffff) return SK_ERR_INVALID;
	lay->in_block = (int32_t)in; lay->out_block = (int32_t)out;
	return SK_OK;
}

int sk_fused_pass_blocked_dev(sk_ctx *c, const sk_blocked_layout *lay, const uint8_t *in, uint8_t *out, int64_t n, uint8_t min_baseq, uint64_t *counts)
{
	if (!c) return SK_ERR_INVALID;
	if (!lay) return fail(c, SK_ERR_INVALID, "layout is NULL");
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld is negative", (long long)n);
	if (n == 0) return SK_OK;
	sk_blocked_layout want;
	if (sk_blocked_layout_init(&want, lay->n_mates, lay->stride, lay->bc_stride, lay->flags) != SK_OK || memcmp(&want, lay, sizeof want) != 0)
		return fail(c, SK_ERR_INVALID, "layout was not made by sk_blocked_layout_init");
	if (!in || !out || !aligned16(in) || !aligned16(out)) return fail(c, SK_ERR_INVALID, "in/out must be 16-byte aligned device buffers");
	if (lay->bc_stride > 0) {
		if (!c->have_table) return fail(c, SK_ERR_STATE, "sk_set_barcodes has not been called");
		if (lay->bc_stride < c->L) return fail(c, SK_ERR_INVALID, "bc_stride = %d is below the barcode length %d", lay->bc_stride, c->L);
	}
	if (int r = bind(c)) return r;
	sk::BlockedArgs a;
	memset(&a, 0, sizeof a);
	a.in = in; a.out = out; a.n = n;
	a.n_mates = lay->n_mates; a.stride = lay->stride; a.bc_stride = lay->bc_stride;
	a.in_block = lay->in_block; a.out_block = lay->out_block;
	for (int m = 0; m < 2; m++) {
		a.in_qual[m] = lay->in_qual[m]; a.in_seq[m] = lay->in_seq[m]; a.in_len[m] = lay->in_len[m];
		a.out_seq[m] = lay->out_seq[m]; a.out_lowest_k[m] = lay->out_lowest_k[m];
	}
	a.in_bc = lay->in_bc; a.out_assign = lay->out_assign; a.out_lowest_diff = lay->out_lowest_diff;
	a.out_first_idx = lay->out_first_idx; a.out_last_idx = lay->out_last_idx;
	a.qc = sk::make_qual_consts(min_baseq);
	if (lay->bc_stride > 0) a.table = table_of(c);
	a.counts = counts ? (unsigned long long *)counts : c->d_counts;
	if (!sk::blocked_shape_ok(a)) return fail(c, SK_ERR_INVALID, "shape not served by the tile-blocked pass (stride %d, bc_stride %d, S %d): use sk_fused_pass_dev", lay->stride, lay->bc_stride, c->S);
	SK_HIP(c, sk::launch_tile_blocked(a, c->n_cu, c->stream));
	return SK_OK;
}

// ---- single-operation entry points are thin views of the fused pass ----------------------------------
int sk_demux_assign_dev(sk_ctx *c, const uint8_t *bc, int bc_stride, int64_t n, int32_t *assign, uint8_t *lowest_diff,
                        int16_t *first_idx, int16_t *last_idx, uint64_t *counts)
{
	if (!c) return SK_ERR_INVALID;
	if (!bc && n > 0) return fail(c, SK_ERR_INVALID, "bc is NULL");
	sk_fused_args a;
	memset(&a, 0, sizeof a);
	a.n = n; a.n_mates = 0; a.bc = bc; a.bc_stride = bc_stride; a.assign = assign;
	a.lowest_diff = lowest_diff; a.first_idx = first_idx; a.last_idx = last_idx; a.counts = counts;
	if (n == 0) return SK_OK;
	return sk_fused_pass_dev(c, &a);
}

int sk_demux_assign(sk_ctx *c, const uint8_t *bc, int bc_stride, int64_t n, int32_t *assign, uint8_t *lowest_diff,
                    int16_t *first_idx, int16_t *last_idx)
{
	if (!c) return SK_ERR_INVALID;
	if (!bc && n > 0) return fail(c, SK_ERR_INVALID, "bc is NULL");
	sk_fused_args a;
	memset(&a, 0, sizeof a);
	a.n = n; a.n_mates = 0; a.bc = bc; a.bc_stride = bc_stride; a.assign = assign;
	a.lowest_diff = lowest_diff; a.first_idx = first_idx; a.last_idx = last_idx;
	if (n == 0) return SK_OK;
	return sk_fused_pass(c, &a);
}

static void one_mate(sk_fused_args &a, const uint8_t *seq, const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                     uint8_t min_baseq, uint8_t *out_seq, uint16_t *lowest_k)
{
	memset(&a, 0, sizeof a);
	a.n = n; a.n_mates = 1; a.stride = stride; a.min_baseq = min_baseq;
	a.mate[0].seq = seq; a.mate[0].qual = qual; a.mate[0].len = len; a.mate[0].out_seq = out_seq; a.mate[0].lowest_k = lowest_k;
}

int sk_trim_by_quality_dev(sk_ctx *c, const uint8_t *qual, const uint16_t *len, int stride, int64_t n, uint8_t min_baseq, uint16_t *lowest_k)
{
	if (!c) return SK_ERR_INVALID;
	if (n == 0) return SK_OK;
	if (!lowest_k) return fail(c, SK_ERR_INVALID, "lowest_k is NULL");
	sk_fused_args a;
	one_mate(a, nullptr, qual, len, stride, n, min_baseq, nullptr, lowest_k);
	return sk_fused_pass_dev(c, &a);
}

int sk_trim_by_quality(sk_ctx *c, const uint8_t *qual, const uint16_t *len, int stride, int64_t n, uint8_t min_baseq, uint16_t *lowest_k)
{
	if (!c) return SK_ERR_INVALID;
	if (n == 0) return SK_OK;
	if (!lowest_k) return fail(c, SK_ERR_INVALID, "lowest_k is NULL");
	sk_fused_args a;
	one_mate(a, nullptr, qual, len, stride, n, min_baseq, nullptr, lowest_k);
	return sk_fused_pass(c, &a);
}

int sk_mask_by_quality_dev(sk_ctx *c, const uint8_t *seq, const uint8_t *qual, int stride, int64_t n, uint8_t min_baseq, uint8_t *out_seq)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0 || stride <= 0) return fail(c, SK_ERR_INVALID, "n = %lld, stride = %d", (long long)n, stride);
	if (n == 0) return SK_OK;
	if (!seq || !qual || !out_seq) return fail(c, SK_ERR_INVALID, "seq, qual or out_seq is NULL");
	if (!aligned16(seq) || !aligned16(qual) || !aligned16(out_seq)) return fail(c, SK_ERR_INVALID, "device byte matrices must be 16-byte aligned");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_mask_flat(seq, qual, out_seq, n * (int64_t)stride, sk::make_qual_consts(min_baseq), c->n_cu, c->stream));
	return SK_OK;
}

int sk_mask_by_quality(sk_ctx *c, uint8_t *seq, const uint8_t *qual, const uint16_t *len, int stride, int64_t n, uint8_t min_baseq)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0 || stride <= 0) return fail(c, SK_ERR_INVALID, "n = %lld, stride = %d", (long long)n, stride);
	if (n == 0) return SK_OK;
	if (!seq || !qual) return fail(c, SK_ERR_INVALID, "seq or qual is NULL");
	(void)len;   // pad bytes of the output are unspecified: the whole matrix is one byte stream
	if (int r = bind(c)) return r;
	const int64_t total = n * (int64_t)stride;
	int64_t chunk = (int64_t)pipe_chunk((size_t)16 << 20);
	if (chunk > total) chunk = (total + 15) & ~(int64_t)15;
	ChunkPipe pipe(c);
	if (int r = pipe.begin((size_t)up256((size_t)chunk) * 3)) return r;
	sk::QualConsts qc = sk::make_qual_consts(min_baseq);
	for (int64_t o = 0; o < total; o += chunk, pipe.next()) {
		const int64_t nb = (total - o) < chunk ? (total - o) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *ds = pipe.ws(), *dq = ds + up256((size_t)chunk), *dout = ds + 2 * up256((size_t)chunk);
		SK_HIP(c, hipMemcpyAsync(ds, seq + o, (size_t)nb, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dq, qual + o, (size_t)nb, hipMemcpyHostToDevice, st));
		SK_HIP(c, sk::launch_mask_flat(ds, dq, dout, nb, qc, c->n_cu, st));
		SK_HIP(c, hipMemcpyAsync(seq + o, dout, (size_t)nb, hipMemcpyDeviceToHost, st));
	}
	return pipe.end();
}

// ---- BAM -------------------------------------------------------------------------------------------
int sk_bam_flag_tlen_dev(sk_ctx *c, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                         int64_t n, int32_t max_frag, uint64_t *out)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0 || max_frag < 0) return fail(c, SK_ERR_INVALID, "n = %lld, max_frag = %d", (long long)n, max_frag);
	if (n == 0) return SK_OK;
	if (!flag || !tid || !mtid || !tlen || !out) return fail(c, SK_ERR_INVALID, "NULL column or out");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bam_flag_tlen(flag, tid, mtid, tlen, n, max_frag, (unsigned long long *)out, 1, 1, c->n_cu, c->stream));
	return SK_OK;
}

int sk_bam_flag_tlen(sk_ctx *c, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                     int64_t n, int32_t max_frag, uint64_t counters[3], uint64_t *hist, uint64_t *hist_total)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0 || max_frag < 0) return fail(c, SK_ERR_INVALID, "n = %lld, max_frag = %d", (long long)n, max_frag);
	if (n == 0) return SK_OK;
	if (!flag) return fail(c, SK_ERR_INVALID, "flag is NULL");
	if (hist && (!tid || !mtid || !tlen)) return fail(c, SK_ERR_INVALID, "histogram needs tid, mtid and tlen");
	if (!counters && !hist) return fail(c, SK_ERR_INVALID, "nothing to do");
	if (int r = bind(c)) return r;
	const size_t nout = 4 + (hist ? (size_t)max_frag + 1 : 0);
	int64_t chunk = (int64_t)pipe_chunk((size_t)4 << 20);
	if (chunk > n) chunk = n;
	const size_t cols = up256((size_t)chunk * 2) + 3 * up256((size_t)chunk * 4);
	ChunkPipe pipe(c);
	if (int r = pipe.begin(cols, up256(nout * 8))) return r;
	unsigned long long *dout = (unsigned long long *)pipe.tail();
	SK_HIP(c, hipMemsetAsync(dout, 0, nout * 8, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));                   // both lanes add into dout
	for (int64_t o = 0; o < n; o += chunk, pipe.next()) {
		const int64_t nr = (n - o) < chunk ? (n - o) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *p = pipe.ws();
		uint16_t *dflag = (uint16_t *)p; p += up256((size_t)chunk * 2);
		int32_t *dtid = (int32_t *)p; p += up256((size_t)chunk * 4);
		int32_t *dmtid = (int32_t *)p; p += up256((size_t)chunk * 4);
		int32_t *dtlen = (int32_t *)p;
		SK_HIP(c, hipMemcpyAsync(dflag, flag + o, (size_t)nr * 2, hipMemcpyHostToDevice, st));
		if (hist) {
			SK_HIP(c, hipMemcpyAsync(dtid, tid + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
			SK_HIP(c, hipMemcpyAsync(dmtid, mtid + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
			SK_HIP(c, hipMemcpyAsync(dtlen, tlen + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
		}
		SK_HIP(c, sk::launch_bam_flag_tlen(dflag, dtid, dmtid, dtlen, nr, max_frag, dout, counters ? 1 : 0, hist ? 1 : 0, c->n_cu, st));
	}
	if (int r = pipe.end()) return r;
	std::vector<uint64_t> h(nout);
	SK_HIP(c, hipMemcpy(h.data(), dout, nout * 8, hipMemcpyDeviceToHost));
	if (counters) for (int i = 0; i < 3; i++) counters[i] += h[i];
	if (hist) {
		if (hist_total) *hist_total += h[3];
		for (size_t i = 0; i <= (size_t)max_frag; i++) hist[i] += h[4 + i];
	}
	return SK_OK;
}

// ---- B1 on the device (sk_inflate.hip) -----------------------------------------------------------------------------
int sk_bgzf_inflate_dev(sk_ctx *c, const uint8_t *comp, const sk_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, uint32_t *status, int check_crc)
{
	if (!c) return SK_ERR_INVALID;
	if (n_blocks < 0) return fail(c, SK_ERR_INVALID, "n_blocks = %lld", (long long)n_blocks);
	if (n_blocks == 0) return SK_OK;
	if (!comp || !blocks || !out || !status) return fail(c, SK_ERR_INVALID, "NULL comp, blocks, out or status");
	if (!aligned16(out)) return fail(c, SK_ERR_INVALID, "out must be 16-byte aligned");
	if ((uintptr_t)blocks & 7u) return fail(c, SK_ERR_INVALID, "blocks must be 8-byte aligned");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bgzf_inflate(comp, blocks, n_blocks, out, status, check_crc, c->n_cu, c->stream));
	return SK_OK;
}

int sk_bam_walk_dev(sk_ctx *c, const uint8_t *stream, uint64_t stream_len, const uint64_t *block_end, int64_t n, uint64_t first_record, int32_t n_ref,
                    uint64_t *entry, uint64_t *exit_scratch, uint32_t *nrec_scratch, int max_rounds, int *verified, uint64_t *n_records, int *rounds)
{
	if (!c || !verified) return SK_ERR_INVALID;
	*verified = 0;
	if (n_records) *n_records = 0;
	if (rounds) *rounds = 0;
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (n == 0) { *verified = first_record == stream_len; return SK_OK; }
	if (!stream || !block_end || !entry || !exit_scratch || !nrec_scratch) return fail(c, SK_ERR_INVALID, "NULL stream, block_end, entry or scratch");
	if (int r = bind(c)) return r;
	uint32_t *changed = nrec_scratch + n;
	SK_HIP(c, sk::launch_bam_walk(stream, stream_len, block_end, entry, exit_scratch, nrec_scratch, n, first_record, changed, 0, n_ref, c->stream));
	int r_done = 1;
	for (;; r_done++) {
		SK_HIP(c, hipMemsetAsync(changed, 0, 4, c->stream));
		// (the first round behind the walk: blocks whose entry is not their predecessor's exit guess a record start and walk from it)
		if (r_done == 1) SK_HIP(c, sk::launch_bam_walk(stream, stream_len, block_end, entry, exit_scratch, nrec_scratch, n, first_record, changed, 2, n_ref, c->stream));
		SK_HIP(c, sk::launch_bam_walk(stream, stream_len, block_end, entry, exit_scratch, nrec_scratch, n, first_record, changed, 1, n_ref, c->stream));
		uint32_t ch = 0;
		SK_HIP(c, hipMemcpyAsync(&ch, changed, 4, hipMemcpyDeviceToHost, c->stream));
		SK_HIP(c, hipStreamSynchronize(c->stream));
		if (ch == 0) break;
		if (r_done >= max_rounds) { if (rounds) *rounds = r_done; return SK_OK; }      // did not settle: not verified
	}
	if (rounds) *rounds = r_done;
	// settled: every entry is its predecessor's exit.  The chain is the file's when it also ends where the stream ends (and no walk met a record it could not take)
	std::vector<uint64_t> ex((size_t)n);
	std::vector<uint32_t> nr((size_t)n);
	SK_HIP(c, hipMemcpy(ex.data(), exit_scratch, (size_t)n * 8, hipMemcpyDeviceToHost));
	SK_HIP(c, hipMemcpy(nr.data(), nrec_scratch, (size_t)n * 4, hipMemcpyDeviceToHost));
	uint64_t total = 0;
	for (int64_t i = 0; i < n; i++) {
		if (ex[(size_t)i] >= ~0ull - 2ull) return SK_OK;
		total += nr[(size_t)i];
	}
	const uint64_t last = ex[(size_t)n - 1] < first_record ? first_record : ex[(size_t)n - 1];
	if (last != stream_len) return SK_OK;
	*verified = 1;
	if (n_records) *n_records = total;
	return SK_OK;
}

int sk_bam_walk_reduce_dev(sk_ctx *c, const uint8_t *stream, uint64_t stream_len, const uint64_t *block_end, const uint64_t *entry, int64_t n,
                           int32_t max_frag, int want_counters, int want_hist, uint64_t *out)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0 || max_frag < 0) return fail(c, SK_ERR_INVALID, "n = %lld, max_frag = %d", (long long)n, max_frag);
	if (n == 0) return SK_OK;
	if (!stream || !block_end || !entry || !out) return fail(c, SK_ERR_INVALID, "NULL stream, block_end, entry or out");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bam_walk_reduce(stream, stream_len, block_end, entry, n, max_frag, want_counters, want_hist, (unsigned long long *)out, c->stream));
	return SK_OK;
}


// ---- F2 on the device (sk_deflate.hip) -------------------------------------------------------------------------------
int sk_bgzf_deflate_dev(sk_ctx *c, const uint8_t *in, const sk_deflate_block *blocks, int64_t n_blocks, uint8_t *slots, uint32_t slot_stride,
                        uint32_t *tokens, uint32_t *result, uint32_t *crc)
{
	if (!c) return SK_ERR_INVALID;
	if (n_blocks < 0) return fail(c, SK_ERR_INVALID, "n_blocks = %lld", (long long)n_blocks);
	if (n_blocks == 0) return SK_OK;
	if (!in || !blocks || !slots || !tokens || !result || !crc) return fail(c, SK_ERR_INVALID, "NULL in, blocks, slots, tokens, result or crc");
	if (slot_stride < SK_DEFLATE_SLOT || (slot_stride & 3u) || ((uintptr_t)slots & 3u)) return fail(c, SK_ERR_INVALID, "slot_stride = %u (>= %d, a multiple of 4), slots 4-byte aligned", slot_stride, SK_DEFLATE_SLOT);
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bgzf_deflate(in, blocks, n_blocks, slots, slot_stride, tokens, result, crc, c->n_cu, c->stream));
	return SK_OK;
}

int sk_bgzf_deflate(sk_ctx *c, const uint8_t *in, size_t in_bytes, const sk_deflate_block *blocks, int64_t n_blocks, uint8_t *out, size_t out_cap, uint64_t *out_off)
{
	if (!c) return SK_ERR_INVALID;
	if (n_blocks < 0 || !out_off) return fail(c, SK_ERR_INVALID, "n_blocks = %lld, out_off %p", (long long)n_blocks, (void *)out_off);
	out_off[0] = 0;
	if (n_blocks == 0) return SK_OK;
	if (!in || !blocks || !out) return fail(c, SK_ERR_INVALID, "NULL in, blocks or out");
	for (int64_t i = 0; i < n_blocks; i++)
		if (blocks[i].in_len > SK_DEFLATE_MAX_IN || blocks[i].in_off + blocks[i].in_len > in_bytes)
			return fail(c, SK_ERR_INVALID, "block %lld: %u bytes at %llu of %zu (at most %d a block)", (long long)i, blocks[i].in_len, (unsigned long long)blocks[i].in_off, in_bytes, SK_DEFLATE_MAX_IN);
	if (int r = bind(c)) return r;
	const size_t n = (size_t)n_blocks;
	const size_t b_in = up256(in_bytes + 8), b_blk = up256(n * sizeof(sk_deflate_block)), b_slots = up256(n * (size_t)SK_DEFLATE_SLOT);
	const size_t b_tok = up256(n * (size_t)SK_DEFLATE_MAX_IN * 4), b_res = up256(n * 8), b_crc = up256(n * 4);
	if (int r = ensure_ws(c, b_in + b_blk + b_slots + b_tok + b_res + b_crc)) return r;
	uint8_t *d_in = c->ws, *d_blk = d_in + b_in, *d_slots = d_blk + b_blk, *d_tok = d_slots + b_slots, *d_res = d_tok + b_tok, *d_crc = d_res + b_res;
	// a pinned landing area for the slots (the ctx keeps it)
	if (c->pin_bytes < b_slots) {
		if (c->pin) { SK_HIP(c, hipHostFree(c->pin)); c->pin = nullptr; c->pin_bytes = 0; }
		SK_HIP(c, hipHostMalloc((void **)&c->pin, b_slots + (b_slots >> 2), hipHostMallocDefault));
		c->pin_bytes = b_slots + (b_slots >> 2);
	}
	SK_HIP(c, hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
	SK_HIP(c, hipMemsetAsync(d_in + in_bytes, 0, 8, c->stream));
	SK_HIP(c, hipMemcpyAsync(d_blk, blocks, n * sizeof(sk_deflate_block), hipMemcpyHostToDevice, c->stream));
	SK_HIP(c, sk::launch_bgzf_deflate(d_in, d_blk, n_blocks, d_slots, SK_DEFLATE_SLOT, (uint32_t *)d_tok, (uint32_t *)d_res, (uint32_t *)d_crc, c->n_cu, c->stream));
	std::vector<uint32_t> res(2 * n), crc(n);
	SK_HIP(c, hipMemcpyAsync(res.data(), d_res, n * 8, hipMemcpyDeviceToHost, c->stream));
	SK_HIP(c, hipMemcpyAsync(crc.data(), d_crc, n * 4, hipMemcpyDeviceToHost, c->stream));
	SK_HIP(c, hipMemcpyAsync(c->pin, d_slots, n * (size_t)SK_DEFLATE_SLOT, hipMemcpyDeviceToHost, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));
	static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
	size_t at = 0;
	for (size_t i = 0; i < n; i++) {
		const uint32_t len = blocks[i].in_len, pay = res[2 * i];
		const bool stored = pay >= len + 5u || pay + 26u > SK_DEFLATE_MAX_MEMBER;
		const size_t clen = stored ? (size_t)len + 5 : pay;
		const size_t bsize = 18 + clen + 8;
		if (at + bsize > out_cap) return fail(c, SK_ERR_INVALID, "out_cap = %zu is too small (block %zu ends at %zu)", out_cap, i, at + bsize);
		uint8_t *m = out + at;
		memcpy(m, head, 16);
		m[16] = (uint8_t)((bsize - 1) & 0xff);
		m[17] = (uint8_t)((bsize - 1) >> 8);
		if (stored) {
			m[18] = 1;                                                    // BFINAL = 1, BTYPE = 00
			m[19] = (uint8_t)(len & 0xff); m[20] = (uint8_t)(len >> 8);
			m[21] = (uint8_t)(~len & 0xff); m[22] = (uint8_t)((~len >> 8) & 0xff);
			memcpy(m + 23, in + blocks[i].in_off, len);
		} else {
			memcpy(m + 18, c->pin + i * (size_t)SK_DEFLATE_SLOT, pay);
		}
		uint8_t *t = m + 18 + clen;
		for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc[i] >> (8 * k)); t[4 + k] = (uint8_t)(len >> (8 * k)); }
		at += bsize;
		out_off[i + 1] = at;
	}
	return SK_OK;
}

// ---- f2: sam fragments filter ----------------------------------------------------------------------------------
int sk_bam_fragments_dev(sk_ctx *c, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                         int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep_bits, uint64_t *kept)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (n == 0) return SK_OK;
	if (!flag || !tid || !mtid || !tlen || !keep_bits || !kept) return fail(c, SK_ERR_INVALID, "NULL column or output");
	if (!aligned16(flag) || !aligned16(tid) || !aligned16(mtid) || !aligned16(tlen)) return fail(c, SK_ERR_INVALID, "columns must be 16-byte aligned");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bam_fragments(flag, tid, mtid, tlen, n, min_size, max_size, keep_bits, (unsigned long long *)kept, c->n_cu, c->stream));
	return SK_OK;
}

int sk_bam_fragments(sk_ctx *c, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                     int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep_bits, uint64_t *kept)
{
	if (!c) return SK_ERR_INVALID;
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (n == 0) return SK_OK;
	if (!flag || !tid || !mtid || !tlen || !keep_bits) return fail(c, SK_ERR_INVALID, "NULL column or output");
	if (int r = bind(c)) return r;
	int64_t chunk = (int64_t)pipe_chunk((size_t)4 << 20);
	if (chunk > n) chunk = (n + 7) & ~(int64_t)7;
	const size_t need = up256((size_t)chunk * 2) + 3 * up256((size_t)chunk * 4) + up256((size_t)chunk / 8 + 8);
	ChunkPipe pipe(c);
	if (int r = pipe.begin(need, 256)) return r;
	unsigned long long *dkept = (unsigned long long *)pipe.tail();
	SK_HIP(c, hipMemsetAsync(dkept, 0, 8, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));                   // both lanes add into dkept
	for (int64_t o = 0; o < n; o += chunk, pipe.next()) {         // chunk is a multiple of 8 records: byte-aligned in keep_bits
		const int64_t nr = (n - o) < chunk ? (n - o) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *p = pipe.ws();
		uint16_t *dflag = (uint16_t *)p; p += up256((size_t)chunk * 2);
		int32_t *dtid = (int32_t *)p; p += up256((size_t)chunk * 4);
		int32_t *dmtid = (int32_t *)p; p += up256((size_t)chunk * 4);
		int32_t *dtlen = (int32_t *)p; p += up256((size_t)chunk * 4);
		uint8_t *dbits = p;
		SK_HIP(c, hipMemcpyAsync(dflag, flag + o, (size_t)nr * 2, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dtid, tid + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dmtid, mtid + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dtlen, tlen + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
		SK_HIP(c, sk::launch_bam_fragments(dflag, dtid, dmtid, dtlen, nr, min_size, max_size, dbits, dkept, c->n_cu, st));
		SK_HIP(c, hipMemcpyAsync(keep_bits + o / 8, dbits, (size_t)((nr + 7) / 8), hipMemcpyDeviceToHost, st));
	}
	if (int r = pipe.end()) return r;
	uint64_t k = 0;
	SK_HIP(c, hipMemcpy(&k, dkept, 8, hipMemcpyDeviceToHost));
	if (kept) *kept += k;
	return SK_OK;
}

// ---- f2 (second half): sam count ---------------------------------------------------------------------------------
int sk_count_set_regions(sk_ctx *c, int n_chr, const int32_t *chr_off, const uint32_t *rstart, const uint32_t *rend,
                         const int32_t *ridx, int64_t n_entries, int64_t n_counts)
{
	if (!c) return SK_ERR_INVALID;
	if (n_chr < 0 || n_entries < 0 || n_counts < n_entries || n_counts > 0x7fffffff || !chr_off || (n_entries > 0 && (!rstart || !rend)))
		return fail(c, SK_ERR_INVALID, "sk_count_set_regions: bad arguments");
	const int64_t n_regions = n_entries;
	if (chr_off[0] != 0 || chr_off[n_chr] != n_regions) return fail(c, SK_ERR_INVALID, "chr_off must run from 0 to n_entries");
	for (int k = 0; k < n_chr; k++) if (chr_off[k] > chr_off[k + 1]) return fail(c, SK_ERR_INVALID, "chr_off must not decrease");
	if (int r = bind(c)) return r;
	// sort each reference's regions by start (stable, like :63) and take the running maximum of their ends
	std::vector<int32_t> order((size_t)n_regions);
	for (int64_t i = 0; i < n_regions; i++) order[(size_t)i] = (int32_t)i;
	for (int k = 0; k < n_chr; k++)
		std::stable_sort(order.begin() + chr_off[k], order.begin() + chr_off[k + 1], [&](int32_t x, int32_t y) { return rstart[x] < rstart[y]; });
	std::vector<uint32_t> s((size_t)n_regions), e((size_t)n_regions), pm((size_t)n_regions);
	std::vector<int32_t> ix((size_t)n_regions);
	for (int k = 0; k < n_chr; k++) {
		uint32_t run = 0;
		for (int32_t i = chr_off[k]; i < chr_off[k + 1]; i++) {
			const int32_t o = order[(size_t)i];
			const int32_t orig = ridx ? ridx[o] : o;
			s[(size_t)i] = rstart[o]; e[(size_t)i] = rend[o]; ix[(size_t)i] = orig;
			if (orig < 0 || orig >= n_counts) return fail(c, SK_ERR_INVALID, "ridx out of range");
			run = std::max(run, rend[o]);
			pm[(size_t)i] = run;
		}
	}
	SK_HIP(c, hipStreamSynchronize(c->stream));
	if (c->d_cnt) { SK_HIP(c, hipFree(c->d_cnt)); c->d_cnt = nullptr; }
	const size_t b_off = up256((size_t)(n_chr + 1) * 4), b_reg = up256((size_t)n_regions * 4 + 4), b_cnt = up256((size_t)n_counts * 4 + 4);
	hipError_t he = hipMalloc((void **)&c->d_cnt, b_off + 4 * b_reg + b_cnt);
	if (he != hipSuccess) { c->d_cnt = nullptr; return fail(c, SK_ERR_NOMEM, "region tables: %s", hipGetErrorString(he)); }
	uint8_t *p = c->d_cnt;
	c->d_chr_off = (int32_t *)p; p += b_off;
	c->d_rstart = (uint32_t *)p; p += b_reg;
	c->d_rend = (uint32_t *)p; p += b_reg;
	c->d_rpmax = (uint32_t *)p; p += b_reg;
	c->d_ridx = (int32_t *)p; p += b_reg;
	c->d_region_frags = (uint32_t *)p;
	SK_HIP(c, hipMemcpyAsync(c->d_chr_off, chr_off, (size_t)(n_chr + 1) * 4, hipMemcpyHostToDevice, c->stream));
	if (n_regions) {
		SK_HIP(c, hipMemcpyAsync(c->d_rstart, s.data(), (size_t)n_regions * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, hipMemcpyAsync(c->d_rend, e.data(), (size_t)n_regions * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, hipMemcpyAsync(c->d_rpmax, pm.data(), (size_t)n_regions * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, hipMemcpyAsync(c->d_ridx, ix.data(), (size_t)n_regions * 4, hipMemcpyHostToDevice, c->stream));
	}
	SK_HIP(c, hipMemsetAsync(c->d_region_frags, 0, b_cnt, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));              // the staging vectors go out of scope
	c->cnt_n_chr = n_chr;
	c->cnt_n_regions = n_counts;
	return SK_OK;
}

static int count_args(sk_ctx *c, sk::CountArgs &a, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mtid,
                      const int32_t *pos, const int32_t *mpos, const int32_t *tlen, const int32_t *end_pos, int64_t n,
                      uint8_t min_mapq, uint32_t max_frag_len, int single_end, int center)
{
	if (!c->d_cnt) return fail(c, SK_ERR_INVALID, "sk_count_set_regions has not been called");
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (n > 0 && (!flag || !mapq || !tid || !pos || (single_end ? !end_pos : (!mtid || !mpos || !tlen)))) return fail(c, SK_ERR_INVALID, "NULL column");
	a.flag = flag; a.mapq = mapq; a.tid = tid; a.mtid = mtid; a.pos = pos; a.mpos = mpos; a.tlen = tlen; a.end_pos = end_pos;
	a.n = n; a.min_mapq = min_mapq; a.max_frag_len = max_frag_len; a.single_end = single_end ? 1 : 0; a.center = center ? 1 : 0;
	a.n_chr = c->cnt_n_chr; a.chr_off = c->d_chr_off; a.rstart = c->d_rstart; a.rend = c->d_rend; a.rpmax = c->d_rpmax; a.ridx = c->d_ridx;
	a.counts = c->d_region_frags;
	return SK_OK;
}

int sk_count_add_dev(sk_ctx *c, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mtid, const int32_t *pos,
                     const int32_t *mpos, const int32_t *tlen, const int32_t *end_pos, int64_t n, uint8_t min_mapq, uint32_t max_frag_len,
                     int single_end, int center)
{
	if (!c) return SK_ERR_INVALID;
	sk::CountArgs a;
	if (int r = count_args(c, a, flag, mapq, tid, mtid, pos, mpos, tlen, end_pos, n, min_mapq, max_frag_len, single_end, center)) return r;
	if (n == 0) return SK_OK;
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bam_count(a, c->n_cu, c->stream));
	return SK_OK;
}

int sk_count_add(sk_ctx *c, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mtid, const int32_t *pos,
                 const int32_t *mpos, const int32_t *tlen, const int32_t *end_pos, int64_t n, uint8_t min_mapq, uint32_t max_frag_len,
                 int single_end, int center)
{
	if (!c) return SK_ERR_INVALID;
	sk::CountArgs a;
	if (int r = count_args(c, a, flag, mapq, tid, mtid, pos, mpos, tlen, end_pos, n, min_mapq, max_frag_len, single_end, center)) return r;
	if (n == 0) return SK_OK;
	if (int r = bind(c)) return r;
	int64_t chunk = (int64_t)pipe_chunk((size_t)2 << 20);
	if (chunk > n) chunk = n;
	const size_t b2 = up256((size_t)chunk * 2), b1 = up256((size_t)chunk), b4 = up256((size_t)chunk * 4);
	ChunkPipe pipe(c);
	if (int r = pipe.begin(b2 + b1 + 6 * b4)) return r;
	const int32_t *hcol[6] = {tid, mtid, pos, mpos, tlen, end_pos};
	for (int64_t o = 0; o < n; o += chunk, pipe.next()) {
		const int64_t nr = (n - o) < chunk ? (n - o) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *p = pipe.ws();
		uint16_t *dflag = (uint16_t *)p; p += b2;
		uint8_t *dmapq = p; p += b1;
		int32_t *dcol[6];
		for (int k = 0; k < 6; k++) { dcol[k] = (int32_t *)p; p += b4; }
		SK_HIP(c, hipMemcpyAsync(dflag, flag + o, (size_t)nr * 2, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dmapq, mapq + o, (size_t)nr, hipMemcpyHostToDevice, st));
		for (int k = 0; k < 6; k++)
			if (hcol[k]) SK_HIP(c, hipMemcpyAsync(dcol[k], hcol[k] + o, (size_t)nr * 4, hipMemcpyHostToDevice, st));
		a.flag = dflag; a.mapq = dmapq; a.tid = dcol[0]; a.mtid = dcol[1]; a.pos = dcol[2]; a.mpos = dcol[3]; a.tlen = dcol[4]; a.end_pos = dcol[5];
		a.n = nr;
		SK_HIP(c, sk::launch_bam_count(a, c->n_cu, st));
	}
	return pipe.end();
}

int sk_count_get(sk_ctx *c, uint32_t *region_frags)
{
	if (!c) return SK_ERR_INVALID;
	if (!c->d_cnt) return fail(c, SK_ERR_INVALID, "sk_count_set_regions has not been called");
	if (c->cnt_n_regions > 0 && !region_frags) return fail(c, SK_ERR_INVALID, "region_frags is NULL");
	if (int r = bind(c)) return r;
	SK_HIP(c, hipStreamSynchronize(c->stream));
	if (c->cnt_n_regions) SK_HIP(c, hipMemcpy(region_frags, c->d_region_frags, (size_t)c->cnt_n_regions * 4, hipMemcpyDeviceToHost));
	return SK_OK;
}

// ---- fasta gc content -------------------------------------------------------------------------------------------
int sk_gc_set_genome(sk_ctx *c, const uint8_t *genome, int64_t genome_len)
{
	if (!c) return SK_ERR_INVALID;
	if (genome_len < 0 || (genome_len > 0 && !genome)) return fail(c, SK_ERR_INVALID, "sk_gc_set_genome: bad arguments");
	if (int r = bind(c)) return r;
	SK_HIP(c, hipStreamSynchronize(c->stream));
	if (c->d_genome) { SK_HIP(c, hipFree(c->d_genome)); c->d_genome = nullptr; c->genome_len = 0; }
	hipError_t e = hipMalloc((void **)&c->d_genome, (size_t)genome_len + 32);
	if (e != hipSuccess) { c->d_genome = nullptr; return fail(c, SK_ERR_NOMEM, "genome of %lld bytes: %s", (long long)genome_len, hipGetErrorString(e)); }
	if (genome_len) SK_HIP(c, hipMemcpy(c->d_genome, genome, (size_t)genome_len, hipMemcpyHostToDevice));
	SK_HIP(c, hipDeviceSynchronize());     // the ctx streams are non-blocking: make the upload visible to them (as sk_set_barcodes does)
	c->genome_len = genome_len;
	return SK_OK;
}

int sk_gc_count(sk_ctx *c, const int64_t *start, const int64_t *len, int64_t n_regions, uint64_t *gc, uint64_t *total)
{
	if (!c) return SK_ERR_INVALID;
	if (!c->d_genome) return fail(c, SK_ERR_INVALID, "sk_gc_set_genome has not been called");
	if (n_regions < 0 || n_regions > 0x7fffffff) return fail(c, SK_ERR_INVALID, "n_regions = %lld", (long long)n_regions);
	if (n_regions == 0) return SK_OK;
	if (!start || !len || !gc || !total) return fail(c, SK_ERR_INVALID, "NULL argument");
	if (int r = bind(c)) return r;
	// regions -> segments of at most 64 KiB, one wave each
	const int64_t kSeg = 64 * 1024;
	std::vector<int64_t> sstart;
	std::vector<int32_t> slen, sreg;
	for (int64_t i = 0; i < n_regions; i++) {
		if (start[i] < 0 || len[i] < 0 || start[i] > c->genome_len || len[i] > c->genome_len - start[i])
			return fail(c, SK_ERR_INVALID, "region %lld = [%lld, +%lld) leaves the genome (%lld bytes)", (long long)i, (long long)start[i], (long long)len[i], (long long)c->genome_len);
		for (int64_t o = 0; o < len[i]; o += kSeg) {
			sstart.push_back(start[i] + o);
			slen.push_back((int32_t)std::min<int64_t>(kSeg, len[i] - o));
			sreg.push_back((int32_t)i);
		}
	}
	const size_t ns = sstart.size();
	const size_t b_out = up256((size_t)n_regions * 16), b_s = up256(ns * 8 + 8), b_l = up256(ns * 4 + 4);
	if (int r = ensure_ws(c, b_out + b_s + 2 * b_l)) return r;
	unsigned long long *dout = (unsigned long long *)c->ws;
	int64_t *dstart = (int64_t *)(c->ws + b_out);
	int32_t *dlen = (int32_t *)(c->ws + b_out + b_s), *dreg = (int32_t *)(c->ws + b_out + b_s + b_l);
	SK_HIP(c, hipMemsetAsync(dout, 0, (size_t)n_regions * 16, c->stream));
	if (ns) {
		SK_HIP(c, hipMemcpyAsync(dstart, sstart.data(), ns * 8, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, hipMemcpyAsync(dlen, slen.data(), ns * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, hipMemcpyAsync(dreg, sreg.data(), ns * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, sk::launch_gc_count(c->d_genome, dstart, dlen, dreg, (int64_t)ns, dout, c->n_cu, c->stream));
	}
	std::vector<uint64_t> h((size_t)n_regions * 2);
	SK_HIP(c, hipMemcpyAsync(h.data(), dout, (size_t)n_regions * 16, hipMemcpyDeviceToHost, c->stream));
	SK_HIP(c, hipStreamSynchronize(c->stream));
	for (int64_t i = 0; i < n_regions; i++) { gc[i] = h[(size_t)2 * i]; total[i] = h[(size_t)2 * i + 1]; }
	return SK_OK;
}

// ---- f4: sam to fastq sequence() ---------------------------------------------------------------------------------
static int bam_sequence_check(sk_ctx *c, const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride, const uint16_t *flag, int64_t n,
                              uint8_t *out)
{
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (stride < 4 || (stride & 3) || seq4_stride < 4 || (seq4_stride & 3) || 2 * (int64_t)seq4_stride < stride || stride > 65532)
		return fail(c, SK_ERR_INVALID, "stride = %d, seq4_stride = %d: multiples of 4 with 2*seq4_stride >= stride wanted", stride, seq4_stride);
	if (n > 0 && (!seq4 || !qual || !flag || !out)) return fail(c, SK_ERR_INVALID, "NULL matrix or column");
	return SK_OK;
}

int sk_bam_sequence_dev(sk_ctx *c, const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride, const uint16_t *len,
                        const uint16_t *flag, int64_t n, uint8_t min_baseq, uint8_t *out)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bam_sequence_check(c, seq4, seq4_stride, qual, stride, flag, n, out)) return r;
	if (n == 0) return SK_OK;
	if (((uintptr_t)seq4 | (uintptr_t)qual | (uintptr_t)out) & 3u) return fail(c, SK_ERR_INVALID, "matrices must be 4-byte aligned");
	if (int r = bind(c)) return r;
	SK_HIP(c, sk::launch_bam_sequence(seq4, seq4_stride, qual, stride, len, flag, n, min_baseq, out, c->n_cu, c->stream));
	return SK_OK;
}

int sk_bam_sequence(sk_ctx *c, const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride, const uint16_t *len,
                    const uint16_t *flag, int64_t n, uint8_t min_baseq, uint8_t *out)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bam_sequence_check(c, seq4, seq4_stride, qual, stride, flag, n, out)) return r;
	if (n == 0) return SK_OK;
	if (int r = bind(c)) return r;
	const size_t per_row = (size_t)seq4_stride + 2 * (size_t)stride + 4;
	int64_t chunk = (int64_t)(pipe_chunk(kPipeChunkBytes) / per_row);
	if (chunk < 1) chunk = 1;
	if (chunk > n) chunk = n;
	const size_t b_seq = up256((size_t)chunk * seq4_stride), b_q = up256((size_t)chunk * stride), b_col = up256((size_t)chunk * 2);
	ChunkPipe pipe(c);
	if (int r = pipe.begin(b_seq + 2 * b_q + 2 * b_col)) return r;
	for (int64_t o = 0; o < n; o += chunk, pipe.next()) {
		const int64_t nr = (n - o) < chunk ? (n - o) : chunk;
		hipStream_t st = pipe.st();
		uint8_t *dseq = pipe.ws(), *dq = dseq + b_seq, *dout = dq + b_q;
		uint16_t *dlen = (uint16_t *)(dout + b_q), *dflag = (uint16_t *)(dout + b_q + b_col);
		SK_HIP(c, hipMemcpyAsync(dseq, seq4 + o * (int64_t)seq4_stride, (size_t)nr * seq4_stride, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dq, qual + o * (int64_t)stride, (size_t)nr * stride, hipMemcpyHostToDevice, st));
		if (len) SK_HIP(c, hipMemcpyAsync(dlen, len + o, (size_t)nr * 2, hipMemcpyHostToDevice, st));
		SK_HIP(c, hipMemcpyAsync(dflag, flag + o, (size_t)nr * 2, hipMemcpyHostToDevice, st));
		SK_HIP(c, sk::launch_bam_sequence(dseq, seq4_stride, dq, stride, len ? dlen : nullptr, dflag, nr, min_baseq, dout, c->n_cu, st));
		SK_HIP(c, hipMemcpyAsync(out + o * (int64_t)stride, dout, (size_t)nr * stride, hipMemcpyDeviceToHost, st));
	}
	return pipe.end();
}

// ---- f3: barcode census ----------------------------------------------------------------------------------------
static_assert(sizeof(sk_census_entry) == sizeof(sk::CensusEntry), "sk_census_entry layout");

static int census_ready(sk_ctx *c)
{
	if (c->census) return SK_OK;
	SK_HIP(c, sk::census_create(&c->census, c->stream));
	return SK_OK;
}

int sk_census_reset(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (!c->census) return census_ready(c);
	SK_HIP(c, sk::census_reset(c->census, c->n_cu, c->stream));
	return SK_OK;
}

static int census_check(sk_ctx *c, const uint8_t *bc, int bc_stride, int L, int64_t n)
{
	if (n < 0) return fail(c, SK_ERR_INVALID, "n = %lld", (long long)n);
	if (L < 0 || L > sk::kMaxCensusLen) return fail(c, SK_ERR_INVALID, "census barcode length %d (0..%d)", L, sk::kMaxCensusLen);
	if (bc_stride < L || bc_stride < 1 || bc_stride > 64) return fail(c, SK_ERR_INVALID, "bc_stride = %d (1..64), L = %d", bc_stride, L);
	if (n > 0 && !bc) return fail(c, SK_ERR_INVALID, "bc is NULL");
	return SK_OK;
}

int sk_census_add_dev(sk_ctx *c, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = census_check(c, bc, bc_stride, L, n)) return r;
	if (n == 0) return SK_OK;
	if (!aligned16(bc)) return fail(c, SK_ERR_INVALID, "bc must be 16-byte aligned");
	if (int r = bind(c)) return r;
	if (int r = census_ready(c)) return r;
	SK_HIP(c, sk::census_add(c->census, bc, bc_stride, L, n, assign, row_base, c->n_cu, c->stream));
	return SK_OK;
}

int sk_census_add(sk_ctx *c, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = census_check(c, bc, bc_stride, L, n)) return r;
	if (n == 0) return SK_OK;
	if (int r = bind(c)) return r;
	if (int r = census_ready(c)) return r;
	int64_t chunk = (int64_t)(pipe_chunk((size_t)64 << 20) / (size_t)(bc_stride + 4));
	if (chunk > n) chunk = n;
	if (int r = ensure_ws(c, up256((size_t)chunk * bc_stride) + up256((size_t)chunk * 4))) return r;
	uint8_t *dbc = c->ws;
	int32_t *dassign = (int32_t *)(c->ws + up256((size_t)chunk * bc_stride));
	for (int64_t o = 0; o < n; o += chunk) {
		const int64_t nr = (n - o) < chunk ? (n - o) : chunk;
		SK_HIP(c, hipMemcpyAsync(dbc, bc + o * (int64_t)bc_stride, (size_t)nr * bc_stride, hipMemcpyHostToDevice, c->stream));
		if (assign) SK_HIP(c, hipMemcpyAsync(dassign, assign + o, (size_t)nr * 4, hipMemcpyHostToDevice, c->stream));
		SK_HIP(c, sk::census_add(c->census, dbc, bc_stride, L, nr, assign ? dassign : nullptr, row_base + o, c->n_cu, c->stream));
		SK_HIP(c, hipStreamSynchronize(c->stream));
	}
	return SK_OK;
}

int sk_census_stats(sk_ctx *c, uint64_t stats[4])
{
	if (!c || !stats) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (int r = census_ready(c)) return r;
	uint64_t s[4];
	SK_HIP(c, sk::census_stats(c->census, s, c->stream));
	if (s[3] != 0) return fail(c, SK_ERR_HIP, "census table overflow (%llu rows lost)", (unsigned long long)s[3]);
	stats[0] = s[0]; stats[1] = s[1]; stats[2] = s[2]; stats[3] = sk::census_slots(c->census);
	return SK_OK;
}

int sk_census_count_hist(sk_ctx *c, uint64_t hist[64])
{
	if (!c || !hist) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (int r = census_ready(c)) return r;
	SK_HIP(c, sk::census_count_hist(c->census, hist, c->n_cu, c->stream));
	return SK_OK;
}

int sk_census_entries(sk_ctx *c, uint64_t min_count, sk_census_entry *out, uint64_t cap, uint64_t *total)
{
	if (!c || !total || (cap && !out)) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	if (int r = census_ready(c)) return r;
	SK_HIP(c, sk::census_entries(c->census, min_count, reinterpret_cast<sk::CensusEntry *>(out), cap, total, c->n_cu, c->stream));
	return SK_OK;
}

// ---- timing ----------------------------------------------------------------------------------------
int sk_timer_start(sk_ctx *c)
{
	if (!c) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	SK_HIP(c, hipEventRecord(c->ev0, c->stream));
	return SK_OK;
}

int sk_timer_stop(sk_ctx *c, float *ms)
{
	if (!c || !ms) return SK_ERR_INVALID;
	if (int r = bind(c)) return r;
	SK_HIP(c, hipEventRecord(c->ev1, c->stream));
	SK_HIP(c, hipEventSynchronize(c->ev1));
	SK_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
	return SK_OK;
}

}  // extern "C"
