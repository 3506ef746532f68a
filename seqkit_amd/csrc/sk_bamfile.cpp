// sk_bamfile.cpp — sk_bam_file_reduce (include/seqkit_hip.h): `sam statistics` / `sam fragment lengths` over a BAM FILE with
// the device doing what htslib does for the reference (src/common.rs:121-157): inflate every BGZF block, check its CRC-32, walk
// the records — and then the reduction (src/sam_statistics.rs:63-69, src/sam_fragment_lengths.rs:29-43).
//
// The host's part is what only it can do: read the file (a few threads pread it into pinned buffers, in order), ship the
// COMPRESSED bytes, and follow the chain of BGZF headers — 18 bytes per block that say where the next one begins (BSIZE) —
// reading, next to each, the trailer's CRC32 and ISIZE: that is the block table the inflate kernel takes.  Blocks are inflated
// in batches as their bytes arrive (the copy of the next chunk runs under the kernel of the last), into ONE buffer that holds
// the whole inflated stream: records may straddle blocks as they like, the walk sees a plain byte stream.
//
// Nothing here decides that a file is bad.  Whatever is not a regular, complete BGZF file whose every block inflates (on the
// device, or — the blocks the device gave up — with zlib here) to the size and CRC its trailer states, and whose records
// chain from the end of the header exactly to the end of the stream, is left to the caller's record-at-a-time reader
// (*handled = 0), which produces what the reference would.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/seqkit_hip.h"
#include "sk_internal.h"

namespace {

struct Cleanup {                                 // frees what was allocated, whichever way the function is left
	std::vector<void *> dev, pinned;
	std::vector<hipEvent_t> events;
	std::vector<hipStream_t> streams, wait_for;
	int fd = -1;
	~Cleanup()
	{
		// (the big buffers stay with the ctx: nothing of this call may still be running on them when the next one starts)
		for (hipStream_t s : wait_for) (void)hipStreamSynchronize(s);
		for (hipStream_t s : streams) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
		for (void *p : dev) if (p) (void)hipFree(p);
		for (void *p : pinned) if (p) (void)hipHostFree(p);
		for (hipEvent_t e : events) (void)hipEventDestroy(e);
		if (fd >= 0) close(fd);
	}
};

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool pread_full(int fd, uint8_t *dst, size_t n, uint64_t off)
{
	size_t got = 0;
	while (got < n) {
		const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
		if (r < 0) { if (errno == EINTR) continue; return false; }
		if (r == 0) return false;
		got += (size_t)r;
	}
	return true;
}

// [off, off + n) of the file into dst, by `threads` threads
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

// not this path's file: say at which check (info[5] = -check) and leave it to the caller's reader
#define BF_LEAVE(code)                                                                                                  \
	do {                                                                                                                \
		if (info) info[5] = -(double)(code);                                                                            \
		return SK_OK;                                                                                                   \
	} while (0)

// The inflated stream's room: its size is known only when the file's last trailer has been read, and six times the file — what a
// well-compressed BAM needs — is 22 GB for a 3.6 GB file of which 5.5 are used.  Memory of that size given back and taken again is
// what the next call, or the next PROCESS, then waits behind (tools/r06/stall_exp.sh).  So the range is only RESERVED (virtual
// addresses), and physical memory is mapped into it piece by piece as the inflater's frontier moves (hipMemCreate / hipMemMap): what a
// file takes is what it inflates to.  The mapping stays with the ctx.  Where the runtime refuses any of this, plain hipMalloc serves.
struct OutRange {
	uint8_t *va = nullptr;
	size_t reserved = 0, mapped = 0, gran = 0, piece = 0;
	int device = 0;
	std::vector<hipMemGenericAllocationHandle_t> handles;
	bool reserve(size_t bytes, int dev)
	{
		hipMemAllocationProp prop{};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = dev;
		size_t g = 0;
		if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) { (void)hipGetLastError(); return false; }
		gran = g;
		piece = (((size_t)512 << 20) + g - 1) / g * g;
		const size_t want = (bytes + piece - 1) / piece * piece;
		void *p = nullptr;
		if (hipMemAddressReserve(&p, want, 0, nullptr, 0) != hipSuccess || !p) { (void)hipGetLastError(); return false; }
		va = (uint8_t *)p; reserved = want; mapped = 0; device = dev;
		return true;
	}
	bool ensure(size_t bytes)                                            // [0, bytes) is backed by memory
	{
		while (mapped < bytes) {
			if (mapped + piece > reserved) return false;
			hipMemAllocationProp prop{};
			prop.type = hipMemAllocationTypePinned;
			prop.location.type = hipMemLocationTypeDevice;
			prop.location.id = device;
			hipMemGenericAllocationHandle_t h;
			if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
			if (hipMemMap(va + mapped, piece, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
			hipMemAccessDesc acc{};
			acc.location.type = hipMemLocationTypeDevice;
			acc.location.id = device;
			acc.flags = hipMemAccessFlagsProtReadWrite;
			if (hipMemSetAccess(va + mapped, piece, &acc, 1) != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap(va + mapped, piece); (void)hipMemRelease(h); return false; }
			handles.push_back(h);
			mapped += piece;
		}
		return true;
	}
	void release()
	{
		for (size_t i = 0; i < handles.size(); i++) { (void)hipMemUnmap(va + i * piece, piece); (void)hipMemRelease(handles[i]); }
		handles.clear();
		if (va) (void)hipMemAddressFree(va, reserved);
		va = nullptr; reserved = mapped = 0;
	}
	static void destroy(void *p) { OutRange *r = (OutRange *)p; r->release(); delete r; }
};

// (mapping a piece takes ~12 ms — the driver hands out cleared memory —: a thread of its own maps ahead of the frontier while the
// caller's thread reads the file)
struct Mapper {
	OutRange *r = nullptr;
	std::thread th;
	std::mutex m;
	std::condition_variable cv;
	size_t want = 0, have = 0;
	bool stop = false, failed = false;
	void start(OutRange *range)
	{
		r = range; have = r->mapped;
		th = std::thread([this] {
			(void)hipSetDevice(r->device);
			std::unique_lock<std::mutex> lk(m);
			for (;;) {
				cv.wait(lk, [this] { return stop || want > have; });
				if (stop) return;
				const size_t next = have + 1;
				lk.unlock();
				const bool ok = r->ensure(next);                             // one piece
				lk.lock();
				if (!ok) { failed = true; cv.notify_all(); return; }
				have = r->mapped;
				cv.notify_all();
			}
		});
	}
	void ask(size_t bytes) { std::lock_guard<std::mutex> lk(m); if (bytes > want) { want = std::min(bytes, r->reserved); cv.notify_all(); } }
	bool wait_for(size_t bytes) { std::unique_lock<std::mutex> lk(m); if (bytes > want) { want = std::min(bytes, r->reserved); cv.notify_all(); } cv.wait(lk, [&] { return failed || have >= bytes; }); return !failed; }
	~Mapper() { if (th.joinable()) { { std::lock_guard<std::mutex> lk(m); stop = true; cv.notify_all(); } th.join(); } }
};
// what stays with the ctx: the range of the compressed file and the range of the inflated stream
struct Ranges {
	OutRange comp, out;
	static void destroy(void *p) { Ranges *r = (Ranges *)p; r->comp.release(); r->out.release(); delete r; }
};

// The readers: threads that pread the file's chunks, in order, into a ring of page-locked buffers — chunk k into slot k mod R, once the
// copy of chunk k - R out of that slot has been issued and is done — while the caller's thread takes the chunks in order, issues their
// copies and follows the BGZF headers in them.  (A thread per piece of every chunk, started and joined chunk by chunk, left the file
// unread while the headers of a chunk were followed: 3 600 thread starts for a 3.6 GB file.)
struct Readers {
	int fd = -1;
	uint8_t *ring = nullptr;
	size_t chunk = 0;
	uint64_t fsize = 0, n_chunks = 0;
	int R = 0;
	std::vector<hipEvent_t> ev;                  // slot s: the copy out of it
	std::vector<int64_t> ready, copied;          // slot s: the chunk whose bytes are in it / whose copy has been issued (and its headers followed)
	std::vector<std::thread> th;
	std::mutex m;
	std::condition_variable cv;
	uint64_t next = 0;
	bool stop = false, failed = false;
	int device = 0;
	void start(int threads)
	{
		ready.assign((size_t)R, -1); copied.assign((size_t)R, -1);
		for (int t = 0; t < threads; t++) th.emplace_back([this] {
			(void)hipSetDevice(device);
			for (;;) {
				uint64_t k;
				{
					std::unique_lock<std::mutex> lk(m);
					if (stop || next >= n_chunks) return;
					k = next++;
					const int s = (int)(k % (uint64_t)R);
					if (k >= (uint64_t)R) {
						cv.wait(lk, [&] { return stop || copied[(size_t)s] == (int64_t)(k - (uint64_t)R); });
						if (stop) return;
					}
				}
				const int s = (int)(k % (uint64_t)R);
				if (k >= (uint64_t)R) (void)hipEventSynchronize(ev[(size_t)s]);     // (outside the lock: the copy out of the slot)
				const uint64_t off = k * chunk;
				const size_t len = (size_t)std::min<uint64_t>(chunk, fsize - off);
				const bool ok = pread_full(fd, ring + (size_t)s * chunk, len, off);
				std::lock_guard<std::mutex> lk(m);
				if (!ok) failed = true;
				ready[(size_t)s] = (int64_t)k;
				cv.notify_all();
			}
		});
	}
	bool wait_ready(uint64_t k)                   // chunk k is in its slot (false: a read failed)
	{
		std::unique_lock<std::mutex> lk(m);
		const int s = (int)(k % (uint64_t)R);
		cv.wait(lk, [&] { return failed || ready[(size_t)s] == (int64_t)k; });
		return !failed;
	}
	void done_with(uint64_t k)                    // its copy is issued (and recorded in ev), nobody reads the slot any more
	{
		std::lock_guard<std::mutex> lk(m);
		copied[(size_t)(k % (uint64_t)R)] = (int64_t)k;
		cv.notify_all();
	}
	~Readers()
	{
		{ std::lock_guard<std::mutex> lk(m); stop = true; cv.notify_all(); }
		for (auto &t : th) t.join();
	}
};

#define BF_HIP(call)                                                                                                    \
	do {                                                                                                                \
		hipError_t e_ = (call);                                                                                         \
		if (e_ != hipSuccess) return sk::ctx_fail(c, SK_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));               \
	} while (0)

extern "C" int sk_bam_file_reduce(sk_ctx *c, const char *path, int32_t max_frag, uint64_t counters[3], uint64_t *hist, uint64_t *hist_total,
                                  int *handled, double info[8])
{
	if (!c || !path || !handled) return SK_ERR_INVALID;
	*handled = 0;
	if (info) for (int i = 0; i < 8; i++) info[i] = 0.0;
	if (max_frag < 0) return sk::ctx_fail(c, SK_ERR_INVALID, "max_frag = %d", max_frag);
	if (!counters && !hist) return sk::ctx_fail(c, SK_ERR_INVALID, "nothing to do");
	if (int r = sk::ctx_bind(c)) return r;
	Cleanup cl;
	cl.fd = open(path, O_RDONLY);
	if (cl.fd < 0) BF_LEAVE(1);                                        // (the caller's reader says so in the reference's words)
	struct stat sb;
	if (fstat(cl.fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 28) BF_LEAVE(2);
	const uint64_t fsize = (uint64_t)sb.st_size;
	const double t0 = now_ms();
	hipStream_t st = sk::ctx_stream(c), st2 = sk::ctx_stream2(c);
	cl.wait_for = {st, st2};

	// ---- device buffers: the compressed file, and room for the inflated stream (its size is known only when the last
	// trailer has been read: six times the file — a BAM inflates three- to fourfold — or what the device has left).  They STAY WITH
	// THE CTX from one call to the next (sk::ctx_keep; sk_destroy frees them): a process that gave 20 GB back and asked for them again
	// found one call in three waiting 1.3-2.4 s in its reading loop — the copies queue behind what the driver does with memory
	// that changes hands (with a third of the room: none in nine calls; the first call of a process: never).
	enum { kKeepComp = 0, kKeepOut = 1, kKeepPin = 2, kKeepTable = 3, kKeepBlocks = 4, kKeepStatus = 5 };
	int krc = SK_OK;
	uint64_t out_cap = std::max<uint64_t>(fsize * 6, (uint64_t)256 << 20);
	if (const char *ev = getenv("SK_BAMFILE_OUT_FACTOR")) { const int f = atoi(ev); if (f >= 1 && f <= 1100) out_cap = std::max<uint64_t>(fsize * (uint64_t)f, (uint64_t)1 << 20); }
	uint8_t *d_out = nullptr, *d_comp = nullptr;
	Ranges *both = (Ranges *)sk::ctx_ext(c);
	OutRange *range = nullptr, *crange = nullptr;
	if (!getenv("SK_BAMFILE_NO_VMM")) {
		int dev = 0;
		BF_HIP(hipGetDevice(&dev));
		if (!both) { both = new Ranges; sk::ctx_set_ext(c, both, Ranges::destroy); }
		if (both->out.va && both->out.reserved < out_cap + 256) both->out.release();
		if (both->comp.va && both->comp.reserved < fsize + 64) both->comp.release();
		// (the compressed file's range only for big files: mapping costs ~12 ms per 512 MiB whoever asks, a plain allocation of a few GB
		// 6-20 ms — it was an 18 GB one that waited 1.6 s behind another process's exit)
		const bool comp_mapped = fsize > ((uint64_t)8 << 30);
		if (!comp_mapped && both->comp.va) both->comp.release();
		if ((both->out.va || both->out.reserve(out_cap + 256, dev)) && (!comp_mapped || both->comp.va || both->comp.reserve(fsize + 64, dev))) {
			range = &both->out;
			crange = comp_mapped ? &both->comp : nullptr;
		}
	}
	Mapper mapper, cmapper;
	if (range) {
		out_cap = range->reserved - 256;                                 // (a batch asks for its last byte + 128)
		d_out = range->va;
		if (crange) {
			d_comp = crange->va;
			cmapper.start(crange);
			cmapper.ask((size_t)fsize + 64);                              // (all of the file's range, ahead of the readers)
		} else {
			d_comp = (uint8_t *)sk::ctx_keep(c, kKeepComp, fsize + 64, false, &krc);
			if (!d_comp) return krc;
		}
		mapper.start(range);
		mapper.ask(std::min<size_t>((size_t)fsize * 2, range->reserved));   // (a BAM inflates at least that far: on its way before the first byte is read)
	} else if (sk::ctx_kept_bytes(c, kKeepOut) >= out_cap + 64) {
		out_cap = sk::ctx_kept_bytes(c, kKeepOut) - 64;                   // (what an earlier call took: all of it is room)
		d_out = (uint8_t *)sk::ctx_keep(c, kKeepOut, out_cap + 64, false, &krc);
		d_comp = (uint8_t *)sk::ctx_keep(c, kKeepComp, fsize + 64, false, &krc);
		if (!d_comp) return krc;
	} else {
		d_comp = (uint8_t *)sk::ctx_keep(c, kKeepComp, fsize + 64, false, &krc);
		if (!d_comp) return krc;
		size_t free_b = 0, total_b = 0;
		BF_HIP(hipMemGetInfo(&free_b, &total_b));
		out_cap = std::min<uint64_t>(out_cap, (uint64_t)((free_b + sk::ctx_kept_bytes(c, kKeepOut)) * 0.8));
		d_out = (uint8_t *)sk::ctx_keep(c, kKeepOut, out_cap + 64, false, &krc);
		if (!d_out) BF_LEAVE(3);
	}

	// ---- read, ship, follow the headers; inflate batch by batch
	size_t chunk = (size_t)4 << 20;                                      // (a ring of 4 MiB page-locked buffers, one and a half per reader: 48 MiB for 8 readers)
	if (const char *ev = getenv("SK_BAMFILE_CHUNK_LOG2")) { const int lg = atoi(ev); if (lg >= 12 && lg <= 30) chunk = (size_t)1 << lg; }
	int threads = 8;                                                    // (3.6 GB from the page cache: 165-185 ms with 4 readers, 130-155 with 8, the same with 12)
	{ const unsigned hc = std::thread::hardware_concurrency(); if (hc >= 1 && hc < 8) threads = (int)hc; }
	if (const char *ev = getenv("SK_BAMFILE_THREADS")) { const int t = atoi(ev); if (t >= 1 && t <= 64) threads = t; }
	const int kBufs = threads + threads / 2 + 1;
	Readers rd;
	rd.fd = cl.fd; rd.chunk = chunk; rd.fsize = fsize; rd.R = kBufs;
	rd.ring = (uint8_t *)sk::ctx_keep(c, kKeepPin, (size_t)kBufs * chunk, true, &krc);
	if (!rd.ring) return krc;
	BF_HIP(hipGetDevice(&rd.device));
	for (int i = 0; i < kBufs; i++) {
		hipEvent_t e;
		BF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		cl.events.push_back(e);
		rd.ev.push_back(e);
	}
	hipEvent_t ev_batch;
	BF_HIP(hipEventCreateWithFlags(&ev_batch, hipEventDisableTiming));
	cl.events.push_back(ev_batch);

	// the block table (pinned: its batches are copied while the next ones are written) and its device copy: room for blocks of
	// 8 KiB on average — a file of smaller ones is not worth this path
	const size_t tab_cap = (size_t)(fsize / 8192) + 1024;
	struct BlockTable {
		sk_bgzf_block *p = nullptr;
		size_t n = 0;
		size_t size() const { return n; }
		sk_bgzf_block *data() const { return p; }
		sk_bgzf_block &operator[](size_t i) const { return p[i]; }
	} blocks;
	blocks.p = (sk_bgzf_block *)sk::ctx_keep(c, kKeepTable, tab_cap * sizeof(sk_bgzf_block), true, &krc);
	if (!blocks.p) return krc;
	std::vector<uint64_t> bend;
	bend.reserve(tab_cap);
	const double t_alloc_pre = now_ms();
	(void)t_alloc_pre;
	sk_bgzf_block *d_blocks = (sk_bgzf_block *)sk::ctx_keep(c, kKeepBlocks, tab_cap * sizeof(sk_bgzf_block), false, &krc);
	if (!d_blocks) return krc;
	uint32_t *d_status = (uint32_t *)sk::ctx_keep(c, kKeepStatus, tab_cap * sizeof(uint32_t), false, &krc);
	if (!d_status) return krc;

	const double t_alloc = now_ms();
	// A launch is a wave per block, and a CU holds 16 of them: a batch is a whole number of such rounds (what is left over of
	// 8 192 + a chunk's blocks would be a last round that fills a fifth of the chip for as long as a full one takes: 16 ms for a
	// block of literals), and batches alternate between two streams, so that one batch's last waves and the next one's first
	// share the chip.
	const size_t slots = (size_t)sk::ctx_n_cu(c) * 16;
	size_t min_batch = 2 * slots;
	if (const char *ev = getenv("SK_BAMFILE_BATCH")) { const int v = atoi(ev); if (v >= 1) min_batch = (size_t)v; }
	bool whole_rounds = true, two_streams = true;
	if (const char *ev = getenv("SK_BAMFILE_ROUNDS")) whole_rounds = atoi(ev) != 0;
	if (const char *ev = getenv("SK_BAMFILE_STREAMS")) two_streams = atoi(ev) >= 2;
	hipStream_t st_b = st;
	hipEvent_t ev_b = nullptr;
	if (two_streams) {
		BF_HIP(hipStreamCreateWithFlags(&st_b, hipStreamNonBlocking));
		cl.streams.push_back(st_b);
		BF_HIP(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
		cl.events.push_back(ev_b);
	}
	int n_batches = 0;
	uint64_t scan = 0, out_off = 0;                                     // the next header's file offset; bytes of the stream so far
	size_t launched = 0;                                                // blocks handed to the device
	bool eof_block_last = false;
	const uint64_t n_chunks = (fsize + chunk - 1) / chunk;
	uint8_t carry[65536 + 64];                                          // a header or trailer that straddles two chunks is read again (pread: rare, and cached)
	rd.n_chunks = n_chunks;
	rd.start(threads);
	for (uint64_t k = 0; k < n_chunks; k++) {
		const uint64_t c_off = k * chunk;
		const size_t c_len = (size_t)std::min<uint64_t>(chunk, fsize - c_off);
		uint8_t *buf = rd.ring + (size_t)(k % (uint64_t)kBufs) * chunk;
		if (!rd.wait_ready(k)) BF_LEAVE(4);
		if (crange && !cmapper.wait_for((size_t)(c_off + c_len) + 64)) BF_LEAVE(3);
		BF_HIP(hipMemcpyAsync(d_comp + c_off, buf, c_len, hipMemcpyHostToDevice, st2));
		BF_HIP(hipEventRecord(rd.ev[(size_t)(k % (uint64_t)kBufs)], st2));
		// the blocks that are complete with this chunk
		const uint64_t have = c_off + c_len;
		auto bytes = [&](uint64_t off, size_t n) -> const uint8_t * {   // n bytes of the file at off (off + n <= have)
			if (off >= c_off) return buf + (off - c_off);
			if (n > sizeof carry || !pread_full(cl.fd, carry, n, off)) return nullptr;
			return carry;
		};
		const size_t first_new = blocks.size();
		while (scan + 18 <= have) {
			const uint8_t *h = bytes(scan, 18);
			if (!h) BF_LEAVE(5);
			if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) BF_LEAVE(6);          // not a BGZF block: the caller's reader sorts it out
			const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
			if (scan + 12 + xlen > have) break;
			const uint8_t *x = bytes(scan, 12 + xlen);
			if (!x) BF_LEAVE(7);
			size_t bsize = 0;
			for (size_t o = 12; o + 4 <= 12 + xlen;) {
				const size_t slen = (size_t)x[o + 2] | ((size_t)x[o + 3] << 8);
				if (x[o] == 'B' && x[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) { bsize = ((size_t)x[o + 4] | ((size_t)x[o + 5] << 8)) + 1; break; }
				o += 4 + slen;
			}
			if (bsize == 0 || bsize < 12 + xlen + 8) BF_LEAVE(8);
			if (scan + bsize > have) break;                                 // its trailer comes with a later chunk
			const uint8_t *tr = bytes(scan + bsize - 8, 8);
			if (!tr) BF_LEAVE(9);
			sk_bgzf_block b;
			b.in_off = scan + 12 + xlen;
			b.in_len = (uint32_t)(bsize - 12 - xlen - 8);
			b.crc32 = le32(tr);
			b.out_len = le32(tr + 4);
			b.out_off = out_off;
			b.reserved = 0;
			if (b.out_len > 65536u) BF_LEAVE(10);                            // (BGZF: at most 64 KiB per block)
			out_off += b.out_len;
			if (out_off > out_cap) BF_LEAVE(11);                             // inflates further than the room taken: the caller's reader streams it
			if (blocks.n >= tab_cap) BF_LEAVE(12);
			blocks.p[blocks.n++] = b;
			bend.push_back(out_off);
			eof_block_last = b.out_len == 0;
			scan += bsize;
		}
		// a batch of blocks: copied on st2, inflated on st behind the copy.  A launch wants thousands of blocks (a wave per block,
		// 16 waves per CU: 4 096 in flight): the blocks of several chunks go together
		(void)first_new;
		size_t n_new = blocks.size() - launched;
		const bool last = k + 1 == n_chunks;
		if (n_new && (n_new >= min_batch || last)) {
			if (!last && whole_rounds && n_new >= slots) n_new -= n_new % slots;      // (what is left over goes with the next batch)
			const size_t first_new = launched;
			if (range) {
				const size_t upto = (size_t)(blocks[first_new + n_new - 1].out_off + blocks[first_new + n_new - 1].out_len) + 128;
				mapper.ask(upto + ((size_t)3 << 29));                        // (three pieces ahead)
				if (!mapper.wait_for(upto)) BF_LEAVE(3);
			}
			hipStream_t sb = (n_batches & 1) ? st_b : st;
			BF_HIP(hipMemcpyAsync(d_blocks + first_new, blocks.data() + first_new, n_new * sizeof(sk_bgzf_block), hipMemcpyHostToDevice, st2));
			BF_HIP(hipEventRecord(ev_batch, st2));
			BF_HIP(hipStreamWaitEvent(sb, ev_batch, 0));
			BF_HIP(sk::launch_bgzf_inflate(d_comp, d_blocks + first_new, (int64_t)n_new, d_out, d_status + first_new, 1, sk::ctx_n_cu(c), sb));
			launched += n_new;
			n_batches++;
		}
		rd.done_with(k);                                                  // (the chunk's headers have been followed: its slot may be read into again once the copy is through)
	}
	if (two_streams) {                                                   // what follows on st comes behind both streams' batches
		BF_HIP(hipEventRecord(ev_b, st_b));
		BF_HIP(hipStreamWaitEvent(st, ev_b, 0));
	}
	if (scan != fsize) BF_LEAVE(13);                                     // bytes behind the last whole block: a file cut short, or not BGZF to its end
	(void)eof_block_last;                                               // (htslib only warns when the EOF marker is missing; the data are the same)
	if (launched != blocks.size()) BF_LEAVE(13);
	const int64_t nb = (int64_t)blocks.size();
	const uint64_t stream_len = out_off;
	const double t_read = now_ms();
	BF_HIP(hipStreamSynchronize(st2));
	// ---- blocks the device gave up: zlib here
	std::vector<uint32_t> status((size_t)nb);
	BF_HIP(hipMemcpyAsync(status.data(), d_status, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
	{	// (a blocking event: the thread sleeps while the device inflates instead of spinning on the stream)
		hipEvent_t ev_done;
		BF_HIP(hipEventCreateWithFlags(&ev_done, hipEventBlockingSync | hipEventDisableTiming));
		cl.events.push_back(ev_done);
		BF_HIP(hipEventRecord(ev_done, st));
		BF_HIP(hipEventSynchronize(ev_done));
	}
	const double t_inflated = now_ms();
	uint64_t n_host = 0;
	{
		std::vector<uint8_t> cbuf, obuf;
		for (int64_t i = 0; i < nb; i++) {
			if (status[(size_t)i] == 0) continue;
			const sk_bgzf_block &b = blocks[(size_t)i];
			if (getenv("SK_BAMFILE_TRACE")) fprintf(stderr, "sk_bam_file_reduce: block %lld (in %u bytes at %llu, out %u) has status %#x: zlib\n", (long long)i, b.in_len, (unsigned long long)b.in_off, b.out_len, status[(size_t)i]);
			cbuf.resize(b.in_len ? b.in_len : 1);
			obuf.resize(b.out_len ? b.out_len : 1);
			if (b.in_len && !pread_full(cl.fd, cbuf.data(), b.in_len, b.in_off)) BF_LEAVE(14);
			z_stream z;
			memset(&z, 0, sizeof z);
			if (inflateInit2(&z, -15) != Z_OK) BF_LEAVE(15);
			z.next_in = cbuf.data(); z.avail_in = b.in_len;
			z.next_out = obuf.data(); z.avail_out = b.out_len;
			const int zr = inflate(&z, Z_FINISH);
			const bool ok = zr == Z_STREAM_END && z.total_out == b.out_len;
			inflateEnd(&z);
			if (!ok) BF_LEAVE(16);                                           // zlib rejects it too: the caller's reader reports it
			if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), obuf.data(), b.out_len) != b.crc32) BF_LEAVE(17);
			if (b.out_len) BF_HIP(hipMemcpy(d_out + b.out_off, obuf.data(), b.out_len, hipMemcpyHostToDevice));
			n_host++;
		}
	}
	// ---- the BAM header: magic, text, references (SAMv1 §4.2) — where the first record begins
	uint64_t first = 0;
	int32_t n_ref_hdr = -1;
	{
		std::vector<uint8_t> hd;
		size_t want = (size_t)std::min<uint64_t>(stream_len, (uint64_t)1 << 20);
		for (;;) {
			hd.resize(want);
			if (want) BF_HIP(hipMemcpy(hd.data(), d_out, want, hipMemcpyDeviceToHost));
			bool more = false, bad = false;
			auto need = [&](uint64_t end) { if (end > want) { (end > stream_len ? bad : more) = true; return false; } return true; };
			uint64_t o = 0;
			do {
				if (!need(12)) break;
				if (memcmp(hd.data(), "BAM\1", 4) != 0) { bad = true; break; }
				o = 8 + (uint64_t)le32(hd.data() + 4);
				if (!need(o + 4)) break;
				const uint32_t n_ref = le32(hd.data() + o);
				n_ref_hdr = n_ref <= 0x7fffffffu ? (int32_t)n_ref : -1;
				o += 4;
				for (uint32_t r = 0; r < n_ref && !bad && !more; r++) {
					if (!need(o + 4)) break;
					const uint64_t l_name = le32(hd.data() + o);
					if (l_name > (1u << 20)) { bad = true; break; }              // (the caller's reader refuses such a header)
					o += 4 + l_name + 4;
					if (!need(o)) break;
				}
			} while (false);
			if (bad) BF_LEAVE(18);
			if (!more) { first = o; break; }
			if (want >= stream_len) BF_LEAVE(19);
			want = (size_t)std::min<uint64_t>(stream_len, (uint64_t)want * 4);
		}
	}
	const double t_header = now_ms();
	// ---- the records: walk, verify, reduce
	uint64_t *d_bend = nullptr, *d_entry = nullptr, *d_exit = nullptr, *d_red = nullptr;
	uint32_t *d_nrec = nullptr;
	BF_HIP(hipMalloc((void **)&d_bend, (size_t)(nb + 1) * 8)); cl.dev.push_back(d_bend);
	BF_HIP(hipMalloc((void **)&d_entry, (size_t)(nb + 1) * 8)); cl.dev.push_back(d_entry);
	BF_HIP(hipMalloc((void **)&d_exit, (size_t)(nb + 1) * 8)); cl.dev.push_back(d_exit);
	BF_HIP(hipMalloc((void **)&d_nrec, (size_t)(nb + 2) * 4)); cl.dev.push_back(d_nrec);
	const size_t nred = 4 + (size_t)max_frag + 1;
	BF_HIP(hipMalloc((void **)&d_red, nred * 8)); cl.dev.push_back(d_red);
	BF_HIP(hipMemcpyAsync(d_bend, bend.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
	BF_HIP(hipMemsetAsync(d_red, 0, nred * 8, st));
	BF_HIP(hipMemsetAsync(d_out + stream_len, 0, 64, st));               // (the walk reads whole dwords)
	int verified = 0, rounds = 0;
	uint64_t n_records = 0;
	int max_rounds = 64;
	if (const char *ev = getenv("SK_BAMFILE_MAX_ROUNDS")) { const int v = atoi(ev); if (v >= 1) max_rounds = v; }
	if (int r = sk_bam_walk_dev(c, d_out, stream_len, d_bend, nb, first, n_ref_hdr, d_entry, d_exit, d_nrec, max_rounds, &verified, &n_records, &rounds)) return r;
	if (!verified) BF_LEAVE(20);
	const double t_walk = now_ms();
	if (int r = sk_bam_walk_reduce_dev(c, d_out, stream_len, d_bend, d_entry, nb, max_frag, counters ? 1 : 0, hist ? 1 : 0, d_red)) return r;
	std::vector<uint64_t> red(nred);
	BF_HIP(hipMemcpyAsync(red.data(), d_red, nred * 8, hipMemcpyDeviceToHost, st));
	BF_HIP(hipStreamSynchronize(st));
	if (counters) for (int i = 0; i < 3; i++) counters[i] += red[(size_t)i];
	if (hist) {
		if (hist_total) *hist_total += red[3];
		for (size_t i = 0; i <= (size_t)max_frag; i++) hist[i] += red[4 + i];
	}
	*handled = 1;
	if (getenv("SK_BAMFILE_TRACE"))
		fprintf(stderr, "sk_bam_file_reduce: alloc %.1f ms, read + copy + launches %.1f ms, wait for the inflate %.1f ms, host blocks + header %.1f ms, walk %.1f ms, reduce %.1f ms; %lld blocks, %llu by zlib\n",
		        t_alloc - t0, t_read - t_alloc, t_inflated - t_read, t_header - t_inflated, t_walk - t_header, now_ms() - t_walk, (long long)nb, (unsigned long long)n_host);
	if (info) {
		info[0] = (double)fsize; info[1] = (double)stream_len; info[2] = (double)nb; info[3] = (double)n_records;
		info[4] = (double)n_host; info[5] = (double)rounds; info[6] = t_read - t0; info[7] = now_ms() - t_read;
	}
	return SK_OK;
}
