// host_common.h — host-side plumbing of the `fasta` / `sam` binaries that sit above the C-ABI.
//
// Mirrors the reference's src/common.rs for the hot-path commands: the error!/exit(-1) convention
// (:11-16), docopt-style argument grammar (:18-22), FileReader (:83-112: "-" = stdin, "*.gz" = gzip
// stream, strict line reads that demand valid UTF-8) and GzipWriter (:49-81).  Differences that do not
// change any byte a user can observe: gzip streams are read and written in-process with zlib instead of
// through `gunzip`/`gzip` child processes, and output is buffered.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "seqkit_hip.h"

namespace host {

// ---- process exit conventions ------------------------------------------------------------------------
[[noreturn]] void error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));   // "ERROR: ...\n", status 255
[[noreturn]] void panic(const char *what);                                              // a Rust panic: status 101
void at_exit_flush(void (*fn)());                                                       // run before error()/panic() exit
void flush_for_exit();                                   // stdout's buffer, before a main() leaves through _exit

// ---- Rust std text semantics used on the path --------------------------------------------------------
bool utf8_valid(const uint8_t *s, size_t n);             // what BufRead::read_line accepts
unsigned cpu_budget();                                    // CPUs this process may use: affinity mask and cgroup quota, not the machine's count
size_t trim_end_len(const std::string &s);               // str::trim_end(): Unicode White_Space (0x1C..0x1F are NOT)
size_t trim_start_off(const std::string &s);             // offset after leading White_Space
bool is_ascii(const std::string &s);
// regex " BC:[ACGTNacgtn+]+", leftmost-first (src/fasta_demultiplex.rs:38); allow_plus = false is the
// " BC:[ACGTNacgtn]+" of src/fasta_statistics.rs:16
bool find_bc_field(const std::string &h, size_t &start, size_t &end, bool allow_plus = true);
bool find_umi_field(const std::string &h, size_t &start, size_t &end);  // regex " UMI:[^\s]*", leftmost
bool parse_uint(const char *s, uint64_t max, uint64_t &out);            // str::parse::<uN>(): [+]digits, no overflow
std::string fmt_pct(double v);                           // "{:.1}" incl. NaN / inf spellings

// ---- argument grammar ----------------------------------------------------------------------------------
struct Opt { const char *name; bool takes_value; bool present; std::string value; };
// long options only (unique-prefix match, --o=v and --o v, "--" ends options, "-" is a positional)
bool parse_args(int argc, char **argv, int first, std::vector<Opt> &opts, std::vector<std::string> &pos, size_t max_pos);

// ---- line reader -----------------------------------------------------------------------------------------
class BgzfStream;
class LineReader {
public:
	explicit LineReader(const std::string &path);        // exits with the reference's message when it cannot open
	~LineReader();
	LineReader(const LineReader &) = delete;
	LineReader &operator=(const LineReader &) = delete;
	// read_line: clears `line`, reads through '\n' (kept); false at EOF.  Invalid UTF-8 sets bad_utf8() and
	// returns false: the caller finishes the records read so far, then reports the I/O error like the reference.
	bool read_line(std::string &line);
	// The same line without the copy: [p, p + n) lies in the reader's buffer and stays valid until the next call on this
	// reader.  Same rules as read_line (false at EOF or on invalid UTF-8, which sets bad_utf8()).
	bool next_line(const char *&p, size_t &n);
	bool bad_utf8() const { return bad_; }
private:
	bool fill();
	long read_more(uint8_t *dst, size_t n);              // from whichever source this reader has; < 0: read error
	gzFile gz_ = nullptr;
	BgzfStream *bz_ = nullptr;                           // a *.gz input that is BGZF: inflated block-parallel
	int fd_ = -1;
	std::vector<uint8_t> buf_;
	size_t pos_ = 0, end_ = 0;
	bool eof_ = false, bad_ = false;
};

// ---- gzip writer (one per output file) -----------------------------------------------------------------
// Replaces GzipWriter (src/common.rs:49-81: an unbuffered pipe into one `gzip`/`pigz` child per file).  Output is cut
// into blocks that a shared pool of threads deflates into independent gzip members (a valid .gz is any concatenation
// of members; `gunzip`/zlib read them transparently); each file's members are written in submission order.
// Where the members are deflated: on the CPU threads (the default), or on the device (include/seqkit_hip.h:
// sk_bgzf_deflate) — a command that uses the GPU anyway asks for that BEFORE its first writer writes; SEQKIT_GPU_DEFLATE=0 / 1
// overrides either way.  The decompressed streams are the same bytes.
void gz_deflate_on_device(bool on);
class GzWriter {
public:
	explicit GzWriter(const std::string &path);          // "Cannot open file {} for writing."
	~GzWriter();
	GzWriter(const GzWriter &) = delete;
	GzWriter &operator=(const GzWriter &) = delete;
	void write(const char *p, size_t n);
	void write(const std::string &s) { write(s.data(), s.size()); }
	void write(std::string &&s);                         // takes the string over: the copy into the member happens on a pool thread
	void flush_tail();                                   // hand what is buffered to the pool, wait for nothing
	void close();                                        // flush, wait for this file's blocks, close the descriptor
	struct Impl;
private:
	Impl *impl_;
};

// ---- BGZF reader ("BGZF/BAM reader, parallel block inflate") ------------------------------
// A BAM file is a series of BGZF blocks: gzip members of at most 64 KiB whose extra field carries the compressed
// block size, so block boundaries are known without inflating.  One thread reads raw blocks, a few threads inflate
// them (libdeflate through dlopen when present, zlib otherwise) and verify the CRC, and read() hands the bytes out in
// file order.  Input that is gzip but not BGZF is inflated as one zlib stream.  read() is gzread(): the number of
// bytes delivered (< n only at the end of the data, which is also what a file cut inside a block looks like), or -1
// when the data are corrupt.
class BgzfStream {
public:
	explicit BgzfStream(int fd, bool bam = false);       // takes the descriptor; bam: the blocks are walked as BAM records while they are inflated (bam_records)
	~BgzfStream();
	BgzfStream(const BgzfStream &) = delete;
	BgzfStream &operator=(const BgzfStream &) = delete;
	long read(void *dst, size_t n);
	long skip(size_t n);                                 // read() without a destination: same return values, no copy of what is skipped
	// BAM records without a call per record (src/common.rs:121-157 reads them one by one through htslib).  The fixed core
	// of every record that lies wholly inside the current inflated block — from the stream's position on — is appended to
	// `out` and the position moves behind the last of them; 0 = the record at the position does not (it straddles a block,
	// the data end, or its block_size is invalid: read()/skip() deal with that one record and say which).  The walk of a
	// block that BEGINS with a record (htslib flushes a block rather than split a record, so in files it wrote every block
	// does) has been done by the worker thread that inflated it, while the block was in that core's cache; other blocks are
	// walked here.
	struct BamRec { int32_t tid, pos, mtid, mpos, tlen; uint16_t flag; uint8_t mapq, l_read_name; };
	long bam_records(std::vector<BamRec> &out);
	struct Impl;
private:
	Impl *impl_;
};

// Raw DEFLATE of one whole buffer into one whole buffer (a BGZF block) and zlib's CRC-32, both several times faster than
// zlib's streaming inflate() / crc32() (host_inflate.cpp).  inflate_raw: false = the decoder gave up (irregular code,
// bad distance, sizes that do not match) and the caller asks zlib, whose verdict stands; out_len is the exact output size.
bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len, uint32_t (*tail_crc)(uint32_t, const uint8_t *, size_t));

// ---- buffered stdout -----------------------------------------------------------------------------------
class Out {
public:
	void write(const char *p, size_t n);
	void write(const std::string &s) { write(s.data(), s.size()); }
	void flush();
private:
	std::string buf_;
};
Out &out();

// ---- the GPU contexts (created on first use; no CPU fallback) ------------------------------------------
// SEQKIT_GPUS=0,1,... lists the devices a command spreads its blocks over (default: SEQKIT_GPU, else device 0; a
// device may be listed more than once); every listed device gets SEQKIT_CTXS_PER_GPU contexts (default 2), each with
// its own streams and workspace, so that the copies of one block run under the kernel of another.  A context ("slot")
// is used by one thread at a time: a worker holds a GpuLease around its sk_* calls.  Blocks are dealt to the slots
// round-robin; the results are handed on in input order by the callers, whichever slot computed them.
sk_ctx *gpu();                                           // slot 0: set-up calls and single-threaded commands
void gpu_warmup();
double process_start_s();          // CLOCK_MONOTONIC seconds at this library's static initialisation (SEQKIT_PROF lines)                                       // start creating the contexts in the background (silent on failure)
size_t gpu_slots();                                      // number of contexts
sk_ctx *gpu_slot(size_t i);
void gpu_for_each(const std::function<void(sk_ctx *)> &fn);      // every slot, in order (sk_set_barcodes, ...)
class GpuLease {
public:
	GpuLease();                                          // the next slot in round-robin order that is free (waits for one)
	explicit GpuLease(size_t slot);                      // that slot (waits for it): state that lives in one context (census)
	~GpuLease();
	GpuLease(const GpuLease &) = delete;
	GpuLease &operator=(const GpuLease &) = delete;
	sk_ctx *ctx() const { return ctx_; }
private:
	size_t slot_;
	sk_ctx *ctx_;
};

// ---- pinned staging -----------------------------------------------------------------------------------------
// Batch matrices are packed straight into page-locked memory (sk_malloc_pinned), so the library's H2D / D2H copies
// are DMA transfers instead of staged ones.  A worker takes a PinnedArena for the life of one block (pack -> device
// -> emit); arenas are recycled, and an arena only grows.
class PinnedArena {
public:
	explicit PinnedArena(size_t hint_bytes = 0);         // hint: what this use will take in all (the buffer grows before use, not under it)
	~PinnedArena();                                      // back to the pool
	PinnedArena(const PinnedArena &) = delete;
	PinnedArena &operator=(const PinnedArena &) = delete;
	uint8_t *take(size_t bytes);                         // 64-byte aligned, valid until the arena is destroyed; NOT zeroed
	template <class T> T *take_n(size_t n) { return reinterpret_cast<T *>(take(n * sizeof(T))); }
	struct Buf;
private:
	Buf *buf_;
	size_t used_ = 0;
	std::vector<void *> extra_;                          // overflow blocks of this use (freed on release; the main buffer grows for the next use)
};

// ---- block-parallel record pipeline (the ingest half of SURVEY.md §8f f1) ------------------------------
// The input is cut into blocks of whole records (`lines_per_record` lines each; only the last block may end in a
// partial record), blocks are processed by worker threads, and their outputs are written to stdout in input order.
// A block that hits an input error fills `err` (and `err_code`: 255 error!, 101 panic) after the output of the
// records before it; the pipeline writes that output, stops, and raises the error — the same point the
// record-at-a-time reference would have reached.
struct BlockResult { std::string out; std::string err; int err_code = 255; };
using BlockFn = std::function<void(const char *data, size_t n, bool last_block, BlockResult &res)>;
void run_block_pipeline(const std::string &path, int lines_per_record, const BlockFn &fn);

// A block of input bytes.  The memory is not value-initialised, grows by realloc and goes back to a process-wide free
// list when the block dies, so that after the first few blocks a new one lands on pages that are already mapped (a
// std::string grown chunk by chunk cost the demultiplex reader 1.07 of its 1.67 s in resize()).
class Bytes {
public:
	Bytes() = default;
	~Bytes();
	Bytes(Bytes &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), owned_(o.owned_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; o.owned_ = true; }
	Bytes &operator=(Bytes &&o) noexcept;
	Bytes(const Bytes &) = delete;
	Bytes &operator=(const Bytes &) = delete;
	const char *data() const { return p_; }
	char *data() { return p_; }
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	void clear() { n_ = 0; }
	void reserve(size_t cap);                            // keeps the contents; takes a recycled buffer when this one is empty
	void set_size(size_t n) { n_ = n; }                  // n <= capacity: the bytes up to n are the caller's business
	void append(const char *p, size_t n) { reserve(n_ + n); memcpy(p_ + n_, p, n); n_ += n; }
	void borrow(const char *p, size_t n);                // a view of bytes that somebody else keeps alive (a mapped file): read-only
private:
	char *p_ = nullptr;
	size_t n_ = 0, cap_ = 0;
	bool owned_ = true;
};

// A file read as blocks of whole records (`lines_per_record` lines each): next() returns up to `nrec` records; the last
// block of a file may end in a partial record.  Several files are kept in lockstep by asking each for the same nrec.
class RecordBlocks {
public:
	RecordBlocks(const std::string &path, int lines_per_record);
	~RecordBlocks();
	RecordBlocks(const RecordBlocks &) = delete;
	RecordBlocks &operator=(const RecordBlocks &) = delete;
	bool next(size_t nrec, Bytes &blk);                  // false (and blk empty) at end of file
	struct Impl;
private:
	Impl *impl_;
};

// one line of a block: [p, p+n) including its '\n' when present; n == 0 past the end of the block (EOF semantics)
struct Line { const char *p; size_t n; };
class BlockLines {
public:
	BlockLines(const char *data, size_t n) : cur_(data), end_(data + n) {}
	Line next()
	{
		if (cur_ == end_) return {cur_, 0};
		const char *nl = static_cast<const char *>(memchr(cur_, '\n', (size_t)(end_ - cur_)));
		const char *e = nl ? nl + 1 : end_;
		Line l{cur_, (size_t)(e - cur_)};
		cur_ = e;
		return l;
	}
	bool at_end() const { return cur_ == end_; }
private:
	const char *cur_, *end_;
};
size_t trim_end_len(const char *p, size_t n);
bool is_ascii(const char *p, size_t n);

}  // namespace host
