// host_common.cpp — see host_common.h.
#include "host_common.h"

#include <dlfcn.h>
#include <immintrin.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sched.h>
#include <unistd.h>

#include <cerrno>

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <algorithm>
#include <atomic>
#include <deque>
#include <future>
#include <memory>
#include <map>
#include <mutex>
#include <thread>

namespace host {

// hardware_concurrency() counts the machine; a container is often allowed a fraction of it (cgroup cpu.max), and sizing
// the pools for 256 CPUs under a quota of 16 only buys throttling stalls for every thread, the reader's included.
unsigned cpu_budget()
{
	static const unsigned budget = [] {
		unsigned n = std::thread::hardware_concurrency();
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
		long long quota = -1, period = 0;
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "max 100000" or "1600000 100000"
			char q[32];
			if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
			fclose(f);
		} else {
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
		}
		if (quota > 0 && period > 0) { const unsigned c = (unsigned)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
		return n < 1 ? 1u : n;
	}();
	return budget;
}

static void (*g_flush)() = nullptr;
void at_exit_flush(void (*fn)()) { g_flush = fn; }
void flush_for_exit() { out().flush(); }          // (the commands close their own outputs before they return: the registered flush belongs to a command's lifetime)

// error! of src/common.rs:11-16: eprint!("ERROR: "); eprintln!(...); exit(-1)
void error(const char *fmt, ...)
{
	if (g_flush) g_flush();
	out().flush();
	fputs("ERROR: ", stderr);
	va_list ap;
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fputc('\n', stderr);
	fflush(stderr);
	_exit(255);
}

void panic(const char *what)
{
	if (g_flush) g_flush();
	out().flush();
	fprintf(stderr, "thread 'main' panicked: %s\n", what);
	fflush(stderr);
	_exit(101);
}

// ---- UTF-8 / whitespace --------------------------------------------------------------------------------
// length of the all-ASCII prefix, 32 bytes at a time where the CPU has AVX2 (FASTQ is all ASCII: this is the whole cost
// of the UTF-8 check in practice)
__attribute__((target("avx2"))) static size_t ascii_prefix_avx2(const uint8_t *s, size_t n)
{
	size_t i = 0;
	for (; i + 32 <= n; i += 32)
		if (_mm256_movemask_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i))) != 0) break;
	return i;
}

bool utf8_valid(const uint8_t *s, size_t n)
{
	static const bool avx2 = __builtin_cpu_supports("avx2");
	size_t i = avx2 ? ascii_prefix_avx2(s, n) : 0;
	while (i < n) {
		// ASCII runs, eight bytes at a time (FASTQ is all ASCII: this is the whole cost of the check in practice)
		while (i + 8 <= n) {
			uint64_t w;
			memcpy(&w, s + i, 8);
			if (w & 0x8080808080808080ull) break;
			i += 8;
		}
		if (i >= n) break;
		const uint8_t b = s[i];
		if (b < 0x80) { i++; continue; }
		size_t need;
		uint8_t lo = 0x80, hi = 0xBF;
		if (b >= 0xC2 && b <= 0xDF) need = 1;
		else if (b >= 0xE0 && b <= 0xEF) { need = 2; if (b == 0xE0) lo = 0xA0; if (b == 0xED) hi = 0x9F; }
		else if (b >= 0xF0 && b <= 0xF4) { need = 3; if (b == 0xF0) lo = 0x90; if (b == 0xF4) hi = 0x8F; }
		else return false;
		if (i + need >= n) return false;                 // continuation bytes i+1 .. i+need must exist
		if (s[i + 1] < lo || s[i + 1] > hi) return false;
		for (size_t k = 2; k <= need; k++)
			if ((s[i + k] & 0xC0) != 0x80) return false;
		i += need + 1;
	}
	return true;
}

static bool rust_ws(uint32_t c)
{
	return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
	       c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

static uint32_t decode(const uint8_t *p, size_t l)
{
	if (l == 1) return p[0];
	if (l == 2) return ((uint32_t)(p[0] & 0x1F) << 6) | (p[1] & 0x3F);
	if (l == 3) return ((uint32_t)(p[0] & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
	return ((uint32_t)(p[0] & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
}

size_t trim_end_len(const std::string &s)
{
	const uint8_t *p = reinterpret_cast<const uint8_t *>(s.data());
	size_t n = s.size();
	while (n > 0) {
		size_t i = n - 1;
		while (i > 0 && (p[i] & 0xC0) == 0x80 && n - i < 4) i--;
		if (!rust_ws(decode(p + i, n - i))) break;
		n = i;
	}
	return n;
}

size_t trim_start_off(const std::string &s)
{
	const uint8_t *p = reinterpret_cast<const uint8_t *>(s.data());
	size_t off = 0, n = s.size();
	while (off < n) {
		const uint8_t b = p[off];
		size_t l = b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : 4;
		if (l > n - off) l = n - off;
		if (!rust_ws(decode(p + off, l))) break;
		off += l;
	}
	return off;
}

// regex " UMI:[^\s]*" (src/fasta_simplify_read_ids.rs:26), leftmost; \s is Unicode White_Space
bool find_umi_field(const std::string &h, size_t &start, size_t &end)
{
	const size_t at = h.find(" UMI:");
	if (at == std::string::npos) return false;
	const uint8_t *p = reinterpret_cast<const uint8_t *>(h.data());
	size_t e = at + 5;
	const size_t n = h.size();
	while (e < n) {
		const uint8_t b = p[e];
		size_t l = b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : 4;
		if (l > n - e) l = n - e;
		if (rust_ws(decode(p + e, l))) break;
		e += l;
	}
	start = at;
	end = e;
	return true;
}

bool is_ascii(const std::string &s)
{
	for (unsigned char c : s)
		if (c >= 0x80) return false;
	return true;
}

static bool bc_class(char c)
{
	switch (c) {
	case 'A': case 'C': case 'G': case 'T': case 'N': case 'a': case 'c': case 'g': case 't': case 'n': case '+': return true;
	default: return false;
	}
}

// src/fasta_demultiplex.rs:38 — literal " BC:" followed by a greedy run (>= 1) of the class; leftmost match
bool find_bc_field(const std::string &h, size_t &start, size_t &end, bool allow_plus)
{
	const size_t n = h.size();
	auto cls = [allow_plus](char c) { return bc_class(c) && (allow_plus || c != '+'); };
	for (size_t i = 0; i + 5 <= n; i++) {
		if (h[i] == ' ' && h[i + 1] == 'B' && h[i + 2] == 'C' && h[i + 3] == ':' && cls(h[i + 4])) {
			size_t e = i + 5;
			while (e < n && cls(h[e])) e++;
			start = i; end = e;
			return true;
		}
	}
	return false;
}

bool parse_uint(const char *s, uint64_t max, uint64_t &out)
{
	if (*s == '+') s++;
	if (!*s) return false;
	uint64_t v = 0;
	for (; *s; s++) {
		if (*s < '0' || *s > '9') return false;
		const uint64_t d = (uint64_t)(*s - '0');
		if (v > (max - d) / 10) return false;
		v = v * 10 + d;
	}
	out = v;
	return true;
}

std::string fmt_pct(double v)
{
	char buf[64];
	if (std::isnan(v)) return "NaN";
	if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
	snprintf(buf, sizeof buf, "%.1f", v);
	return buf;
}

// ---- docopt grammar ------------------------------------------------------------------------------------
bool parse_args(int argc, char **argv, int first, std::vector<Opt> &opts, std::vector<std::string> &pos, size_t max_pos)
{
	bool only_pos = false;
	for (int i = first; i < argc; i++) {
		const char *a = argv[i];
		if (!only_pos && strcmp(a, "--") == 0) { only_pos = true; continue; }
		if (!only_pos && a[0] == '-' && a[1] == '-') {
			const char *eq = strchr(a, '=');
			const size_t nl = eq ? (size_t)(eq - a) : strlen(a);
			int hit = -1, nh = 0;
			for (size_t k = 0; k < opts.size(); k++) {
				if (strlen(opts[k].name) == nl && strncmp(opts[k].name, a, nl) == 0) { hit = (int)k; nh = 1; break; }
				if (strncmp(opts[k].name, a, nl) == 0) { hit = (int)k; nh++; }
			}
			if (nh != 1) return false;
			Opt &o = opts[hit];
			if (o.takes_value) {
				if (eq) o.value = eq + 1;
				else if (i + 1 < argc) o.value = argv[++i];
				else return false;
			} else if (eq) {
				return false;
			}
			o.present = true;
			continue;
		}
		if (!only_pos && a[0] == '-' && a[1] != 0) return false;      // the grammar has no short options
		if (pos.size() >= max_pos) return false;
		pos.push_back(a);
	}
	return true;
}

// Is the file behind fd BGZF (a gzip header whose extra field has the 'BC' subfield)?  Looked at with pread: the
// descriptor's position stays where it is, and what is not a seekable file is simply not BGZF here.  The outputs of
// this build's GzWriter are, and so is what bgzip writes: such inputs are inflated by BgzfStream's threads instead of
// one zlib stream.
static bool fd_is_bgzf(int fd)
{
	uint8_t h[1024];
	const ssize_t r = pread(fd, h, sizeof h, 0);
	if (r < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
	const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
	if (12 + xlen > (size_t)r) return false;
	for (size_t o = 12; o + 4 <= 12 + xlen;) {
		const size_t slen = (size_t)h[o + 2] | ((size_t)h[o + 3] << 8);
		if (h[o] == 'B' && h[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return true;
		o += 4 + slen;
	}
	return false;
}

// ---- LineReader ------------------------------------------------------------------------------------------
LineReader::LineReader(const std::string &path)
{
	// src/common.rs:88-103: "-" = stdin, "*.gz" = gzip stream, anything else is read as it is
	if (path == "-") {
		fd_ = 0;
	} else {
		fd_ = open(path.c_str(), O_RDONLY);
		if (fd_ < 0) error("Cannot open file %s for reading.", path.c_str());
		if (path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
			if (fd_is_bgzf(fd_)) {
				bz_ = new BgzfStream(fd_);                // takes the descriptor
				fd_ = -1;
			} else {
				gz_ = gzdopen(fd_, "rb");             // inflates every member of the file, like `gunzip -c`
				if (!gz_) error("Cannot start gunzip process.");
				gzbuffer(gz_, 1 << 18);
			}
		}
	}
	buf_.resize(1 << 18);
}

LineReader::~LineReader()
{
	delete bz_;
	if (gz_) gzclose(gz_);
	else if (fd_ > 0) close(fd_);
}

long LineReader::read_more(uint8_t *dst, size_t n)
{
	if (bz_) return bz_->read(dst, n);
	if (gz_) return gzread(gz_, dst, (unsigned)std::min<size_t>(n, 1u << 30));
	ssize_t r;
	do { r = read(fd_, dst, n); } while (r < 0 && errno == EINTR);
	return (long)r;
}

bool LineReader::fill()
{
	if (eof_) return false;
	const long r = read_more(buf_.data(), buf_.size());
	if (r < 0) error("I/O error while reading from file.");
	pos_ = 0;
	end_ = (size_t)r;
	if (r == 0) { eof_ = true; return false; }
	return true;
}

bool LineReader::read_line(std::string &line)
{
	line.clear();
	if (bad_) return false;
	for (;;) {
		if (pos_ == end_ && !fill()) break;
		const uint8_t *p = buf_.data() + pos_;
		const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', end_ - pos_));
		if (nl) {
			line.append(reinterpret_cast<const char *>(p), (size_t)(nl - p) + 1);
			pos_ += (size_t)(nl - p) + 1;
			break;
		}
		line.append(reinterpret_cast<const char *>(p), end_ - pos_);
		pos_ = end_;
	}
	if (!line.empty() && !utf8_valid(reinterpret_cast<const uint8_t *>(line.data()), line.size())) {
		bad_ = true;
		line.clear();
		return false;
	}
	return !line.empty();
}

bool LineReader::next_line(const char *&p, size_t &n)
{
	p = nullptr;
	n = 0;
	if (bad_) return false;
	size_t scanned = 0;                                       // bytes of [pos_, end_) already known to hold no newline
	for (;;) {
		const uint8_t *b = buf_.data() + pos_;
		const uint8_t *nl = end_ - pos_ > scanned ? static_cast<const uint8_t *>(memchr(b + scanned, '\n', end_ - pos_ - scanned)) : nullptr;
		if (nl) { n = (size_t)(nl - b) + 1; break; }
		scanned = end_ - pos_;
		if (eof_) { n = scanned; break; }                     // the last line has no newline
		// the line continues past what is buffered: move its start to the front and read on (a line longer than the
		// buffer makes the buffer grow)
		if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, scanned); pos_ = 0; end_ = scanned; }
		if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
		const long r = read_more(buf_.data() + end_, buf_.size() - end_);
		if (r < 0) error("I/O error while reading from file.");
		if (r == 0) eof_ = true;
		end_ += (size_t)r;
	}
	if (n == 0) return false;
	p = reinterpret_cast<const char *>(buf_.data() + pos_);
	pos_ += n;
	if (!utf8_valid(reinterpret_cast<const uint8_t *>(p), n)) {
		bad_ = true;
		p = nullptr;
		n = 0;
		return false;
	}
	return true;
}

// ---- GzWriter: block-parallel gzip ------------------------------------------------------------------------
namespace {

struct Job { GzWriter::Impl *w; uint64_t seq; std::vector<std::string> parts; };      // one gzip member = the parts, concatenated

struct Pool {
	std::mutex m;
	std::condition_variable cv_job, cv_room;
	std::deque<Job> q;
	std::vector<std::thread> threads;
	size_t in_flight = 0, max_in_flight = 0;
	bool stop = false;
	bool on_gpu = false;               // the members are deflated on the device (run_gpu): one batching thread instead of the CPU threads
	void start();
	void submit(Job &&j);
	void run();
	void run_gpu();
	~Pool();
};

Pool &pool()
{
	static Pool p;
	return p;
}

constexpr size_t kGzBlock = 512u << 10;      // uncompressed bytes per gzip member

}  // namespace

struct GzWriter::Impl {
	int fd = -1;
	std::vector<std::string> parts;    // what the next member will hold; strings handed over whole are kept as they are
	size_t bytes = 0;                  // ... and their total size
	bool open_tail = false;            // the last part was started by write(p, n) and takes more bytes
	uint64_t next_submit = 0;
	void submit();
	// completion side (guarded by m)
	std::mutex m;
	std::condition_variable cv;
	uint64_t next_write = 0;
	std::map<uint64_t, std::string> ready;
	void completed(uint64_t seq, std::string &&comp)
	{
		std::unique_lock<std::mutex> lk(m);
		ready.emplace(seq, std::move(comp));
		// write every block that is next in line (this thread does the I/O for them)
		for (auto it = ready.find(next_write); it != ready.end(); it = ready.find(next_write)) {
			const std::string &c = it->second;
			size_t off = 0;
			while (off < c.size()) {
				const ssize_t w = ::write(fd, c.data() + off, c.size() - off);
				if (w <= 0) break;                   // write errors are ignored like the reference (#![allow(unused_must_use)])
				off += (size_t)w;
			}
			ready.erase(it);
			next_write++;
		}
		cv.notify_all();
	}
};

// libdeflate (when the shared object is on the box; it ships no headers here, so the four entry points are declared by
// hand) compresses the same gzip members ~2-3x faster than zlib at the same level; zlib is the fallback.
struct Deflater {
	void *(*alloc)(int) = nullptr;
	void (*free_)(void *) = nullptr;
	size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;      // raw deflate
	size_t (*bound)(void *, size_t) = nullptr;
	uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
	Deflater()
	{
		if (getenv("SEQKIT_NO_LIBDEFLATE")) return;
		void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
		if (!h) return;
		alloc = reinterpret_cast<void *(*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
		free_ = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_compressor"));
		compress = reinterpret_cast<size_t (*)(void *, const void *, size_t, void *, size_t)>(dlsym(h, "libdeflate_deflate_compress"));
		bound = reinterpret_cast<size_t (*)(void *, size_t)>(dlsym(h, "libdeflate_deflate_compress_bound"));
		crc = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(dlsym(h, "libdeflate_crc32"));
		if (!alloc || !free_ || !compress || !bound || !crc) alloc = nullptr;
	}
};

// The output of one job as BGZF blocks (SAMv1 §4.1: gzip members of at most 64 KiB whose extra field 'BC' carries the
// block size).  Any gzip reader takes them as the members they are; a reader that knows BGZF — htslib's bgzip, this
// build's own BgzfStream behind every *.gz input — finds the block boundaries without inflating and inflates the
// blocks in parallel.
constexpr size_t kBgzfInput = 0xff00;                       // input bytes per block, as htslib cuts them: the block stays under 64 KiB whatever the data
static const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static void bgzf_append_block(std::string &out, const char *in, size_t n)
{
	static const Deflater ld;
	const size_t at = out.size();
	out.resize(at + 18 + n + 1024 + 8);                       // header, deflate at its worst (stored blocks), trailer
	uint8_t *blk = reinterpret_cast<uint8_t *>(&out[at]);
	size_t clen = 0;
	uint32_t crc = 0;
	if (ld.alloc) {
		thread_local void *comp = ld.alloc(6);
		if (comp) {
			clen = ld.compress(comp, in, n, blk + 18, n + 1024);
			crc = ld.crc(0, in, n);
		}
	}
	if (clen == 0) {
		z_stream z;
		memset(&z, 0, sizeof z);
		deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
		z.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(in));
		z.avail_in = (uInt)n;
		z.next_out = blk + 18;
		z.avail_out = (uInt)(n + 1024);
		deflate(&z, Z_FINISH);
		clen = (n + 1024) - z.avail_out;
		deflateEnd(&z);
		crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(in), (uInt)n);
	}
	const size_t bsize = 18 + clen + 8;                         // < 65536: kBgzfInput leaves room for incompressible data
	static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
	memcpy(blk, head, 16);
	blk[16] = (uint8_t)((bsize - 1) & 0xff);
	blk[17] = (uint8_t)((bsize - 1) >> 8);
	uint8_t *t = blk + 18 + clen;
	for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
	out.resize(at + bsize);
}

static std::string gzip_member(const std::string &in)
{
	std::string out;
	out.reserve(in.size() / 3 + 64);
	for (size_t o = 0; o < in.size(); o += kBgzfInput) bgzf_append_block(out, in.data() + o, std::min(kBgzfInput, in.size() - o));
	return out;
}

// ---- BgzfStream ----------------------------------------------------------------------------------------------
struct Inflater {                      // libdeflate's decompressor, when the library is there
	void *(*alloc)() = nullptr;
	void (*free_)(void *) = nullptr;
	int (*inflate)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
	uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
	Inflater()
	{
		if (getenv("SEQKIT_NO_LIBDEFLATE")) return;
		void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
		if (!h) return;
		alloc = reinterpret_cast<void *(*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
		free_ = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
		inflate = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(h, "libdeflate_deflate_decompress"));
		crc = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(dlsym(h, "libdeflate_crc32"));
		if (!alloc || !free_ || !inflate || !crc) alloc = nullptr;
	}
};

struct BgzfStream::Impl {
	struct Block {
		std::vector<uint8_t> comp;         // the whole BGZF block as read ...
		const uint8_t *src = nullptr;      // ... or where it lies in the mapped file (comp stays empty)
		size_t src_len = 0;
		std::vector<uint8_t> data;         // inflated
		size_t cdata_off = 0, cdata_len = 0;
		bool done = false, bad = false;
		// the block walked as BAM records from its first byte, by the worker that inflated it (bam_records)
		std::vector<BamRec> recs;
		size_t recs_end = 0;               // offset behind the last record that lies wholly inside the block
	};
	// The records that lie wholly inside p[start, len): their cores to `out`; returns the offset behind the last of them.
	// Stops at a record that does not fit, and at a block_size that no record can have (the caller's slow path reports it).
	static size_t walk_records(const uint8_t *p, size_t len, size_t start, std::vector<BamRec> &out)
	{
		auto le32 = [](const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
		size_t o = start;
		while (o + 36 <= len) {
			const uint32_t block_size = le32(p + o);
			if (block_size < 32 || (size_t)block_size > len - o - 4) break;
			const uint8_t *c = p + o + 4;
			BamRec r;
			r.tid = (int32_t)le32(c); r.pos = (int32_t)le32(c + 4);
			r.l_read_name = c[8]; r.mapq = c[9];
			r.flag = (uint16_t)(c[14] | (c[15] << 8));
			r.mtid = (int32_t)le32(c + 20); r.mpos = (int32_t)le32(c + 24); r.tlen = (int32_t)le32(c + 28);
			out.push_back(r);
			o += 4 + (size_t)block_size;
		}
		return o;
	}
	int fd;
	bool bgzf = false;                     // decided from the first block header
	// ---- parallel path
	std::mutex mu;
	std::condition_variable cv_work, cv_done, cv_room;
	std::deque<std::shared_ptr<Block>> order;      // blocks in file order (front = next to deliver)
	std::deque<std::shared_ptr<Block>> todo;       // blocks waiting for a worker
	bool eof = false, stop = false, read_error = false;
	std::vector<std::thread> workers;
	std::thread reader;
	std::shared_ptr<Block> cur;
	size_t cur_off = 0;
	size_t max_in_flight = 64;
	// ---- plain gzip path
	z_stream z;
	std::vector<uint8_t> zin;
	bool z_eof = false;
	std::vector<uint8_t> head;             // bytes consumed while sniffing the format

	static bool read_full(int fd, uint8_t *p, size_t n, size_t &got)
	{
		got = 0;
		while (got < n) {
			const ssize_t r = ::read(fd, p + got, n - got);
			if (r < 0) { if (errno == EINTR) continue; return false; }
			if (r == 0) break;
			got += (size_t)r;
		}
		return true;
	}

	// BSIZE of a BGZF block from its gzip header (12 bytes + the XLEN bytes of extra subfields, among which 'B','C'), or 0
	static size_t bgzf_block_size(const uint8_t *h, size_t n, size_t &xlen)
	{
		if (n < 12 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return 0;
		xlen = (size_t)h[10] | ((size_t)h[11] << 8);
		if (12 + xlen > n) return 0;
		for (size_t o = 12; o + 4 <= 12 + xlen;) {
			const size_t slen = (size_t)h[o + 2] | ((size_t)h[o + 3] << 8);
			if (h[o] == 'B' && h[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return ((size_t)h[o + 4] | ((size_t)h[o + 5] << 8)) + 1;
			o += 4 + slen;
		}
		return 0;
	}

	// Read a block's whole gzip header into hdr (on top of what it holds): the 12 fixed bytes, then — when they are a gzip
	// header with FEXTRA — the XLEN bytes of subfields, however many precede 'BC' (RFC 1952; htslib reads such files).
	// false = read error.  A short hdr afterwards means the file ended there.
	static bool read_header(int fd, std::vector<uint8_t> &hdr)
	{
		size_t got = 0;
		if (hdr.size() < 12) {
			const size_t have = hdr.size();
			hdr.resize(12);
			if (!read_full(fd, hdr.data() + have, 12 - have, got)) return false;
			hdr.resize(have + got);
		}
		if (hdr.size() < 12 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return true;
		const size_t want = 12 + ((size_t)hdr[10] | ((size_t)hdr[11] << 8));
		if (hdr.size() < want) {
			const size_t have = hdr.size();
			hdr.resize(want);
			if (!read_full(fd, hdr.data() + have, want - have, got)) return false;
			hdr.resize(have + got);
		}
		return true;
	}

	void reader_main()
	{
		std::vector<uint8_t> hdr(head);                      // the first block's header was read by the constructor
		for (;;) {
			size_t got = 0;
			if (!read_header(fd, hdr)) { fail_read(); return; }
			if (hdr.empty()) break;                          // clean end of file
			size_t xlen = 0;
			const size_t bsize = bgzf_block_size(hdr.data(), hdr.size(), xlen);
			auto b = std::make_shared<Block>();
			if (bsize == 0 || bsize < 12 + xlen + 8) {
				// a cut-off header is the end of the data; anything else that is not a BGZF block is corrupt
				const bool cut = hdr.size() < 12 || (hdr[0] == 31 && hdr[1] == 139 && hdr[2] == 8 && (hdr[3] & 4) && hdr.size() < 12 + (((size_t)hdr[10]) | ((size_t)hdr[11] << 8)));
				if (!cut) { b->bad = true; b->done = true; push(b, false); }
				break;
			}
			const size_t hl = hdr.size();                    // == 12 + xlen
			b->comp.resize(bsize);
			memcpy(b->comp.data(), hdr.data(), hl);
			if (!read_full(fd, b->comp.data() + hl, bsize - hl, got)) { fail_read(); return; }
			if (got != bsize - hl) break;                    // file ends inside a block: the data end before it
			b->cdata_off = 12 + xlen;
			b->cdata_len = bsize - b->cdata_off - 8;
			hdr.clear();
			if (!push(b, true)) return;
		}
		std::lock_guard<std::mutex> lk(mu);
		eof = true;
		cv_done.notify_all();
		cv_work.notify_all();
	}
	// A regular file is mapped instead of read: the one reader thread then only walks the block headers (18 bytes per 64 KiB),
	// and the compressed bytes are touched for the first time by the worker that inflates them — copying them out of the page
	// cache on one thread was what a 256-core host's `sam statistics` waited for (3.6 GB: 0.7 s of its 1.2 s).
	const uint8_t *map = nullptr;
	size_t map_len = 0;
	void reader_main_mapped()
	{
		size_t o = 0;
		while (o < map_len) {
			size_t xlen = 0;
			const size_t left = map_len - o;
			const size_t bsize = bgzf_block_size(map + o, left < 12 + 65535 ? left : 12 + 65535, xlen);
			auto b = std::make_shared<Block>();
			if (bsize == 0 || bsize < 12 + xlen + 8) {
				// a cut-off header is the end of the data; anything else that is not a BGZF block is corrupt
				const uint8_t *h = map + o;
				const bool cut = left < 12 || (h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && left < 12 + (((size_t)h[10]) | ((size_t)h[11] << 8)));
				if (!cut) { b->bad = true; b->done = true; push(b, false); }
				break;
			}
			if (bsize > left) break;                         // file ends inside a block: the data end before it
			b->src = map + o;
			b->src_len = bsize;
			b->cdata_off = 12 + xlen;
			b->cdata_len = bsize - b->cdata_off - 8;
			o += bsize;
			if (!push(b, true)) return;
		}
		std::lock_guard<std::mutex> lk(mu);
		eof = true;
		cv_done.notify_all();
		cv_work.notify_all();
	}
	void fail_read()
	{
		std::lock_guard<std::mutex> lk(mu);
		read_error = true; eof = true;
		cv_done.notify_all();
		cv_work.notify_all();
	}
	bool push(const std::shared_ptr<Block> &b, bool work)
	{
		std::unique_lock<std::mutex> lk(mu);
		cv_room.wait(lk, [&] { return stop || order.size() < max_in_flight; });
		if (stop) return false;
		order.push_back(b);
		if (work) { todo.push_back(b); cv_work.notify_one(); }
		else cv_done.notify_all();
		return true;
	}
	void worker_main()
	{
		static const Inflater ld;
		static const bool no_own_inflate = getenv("SEQKIT_ZLIB_INFLATE") != nullptr;     // tests: zlib's inflate() and crc32() for every block
		void *dec = ld.alloc ? ld.alloc() : nullptr;
		for (;;) {
			std::shared_ptr<Block> b;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_work.wait(lk, [&] { return stop || !todo.empty() || eof; });
				if (stop || (todo.empty() && eof)) { if (dec) ld.free_(dec); return; }
				b = todo.front();
				todo.pop_front();
			}
			const uint8_t *blk = b->src ? b->src : b->comp.data();
			const size_t blk_len = b->src ? b->src_len : b->comp.size();
			const uint8_t *tail = blk + blk_len - 8;
			const uint32_t want_crc = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
			const uint32_t isize = (uint32_t)tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
			bool ok = isize <= (1u << 16);
			if (ok && isize == 0) {
				ok = want_crc == 0;                              // an empty block (the end-of-file marker): nothing to inflate
			} else if (ok) {
				b->data.resize(isize);
				if (dec) {
					size_t out_n = 0;
					ok = ld.inflate(dec, blk + b->cdata_off, b->cdata_len, b->data.data(), isize, &out_n) == 0 && out_n == isize;
					if (ok) ok = ld.crc(0, b->data.data(), isize) == want_crc;
				} else if (!no_own_inflate && inflate_raw(blk + b->cdata_off, b->cdata_len, b->data.data(), isize)) {
					ok = crc32_fast(0, b->data.data(), isize, [](uint32_t c, const uint8_t *p, size_t n) { return (uint32_t)crc32(c, p, (uInt)n); }) == want_crc;
				} else {
					z_stream zs;
					memset(&zs, 0, sizeof zs);
					ok = inflateInit2(&zs, -15) == Z_OK;
					if (ok) {
						zs.next_in = const_cast<uint8_t *>(blk) + b->cdata_off; zs.avail_in = (uInt)b->cdata_len;
						zs.next_out = b->data.data(); zs.avail_out = isize;
						const int rc = ::inflate(&zs, Z_FINISH);
						ok = rc == Z_STREAM_END && zs.avail_out == 0;
						inflateEnd(&zs);
						if (ok) ok = (uint32_t)crc32(crc32(0L, Z_NULL, 0), b->data.data(), isize) == want_crc;
					}
				}
			}
			std::vector<uint8_t>().swap(b->comp);
			if (ok && want_records && b->data.size() >= 36) {      // while the block is in this core's cache
				b->recs.reserve(b->data.size() / 128);
				b->recs_end = walk_records(b->data.data(), b->data.size(), 0, b->recs);
			}
			std::lock_guard<std::mutex> lk(mu);
			b->bad = !ok;
			b->done = true;
			cv_done.notify_all();
		}
	}

	bool failed = false;                   // corrupt data seen: what preceded it has been delivered, every later read is -1
	std::atomic<bool> want_records{false}; // a BAM reader is attached: workers walk every block from its first byte

	// make `cur` the block that holds the stream's position; false at the end of the data or at a corrupt block (nothing consumed)
	bool have_block()
	{
		while (!cur || cur_off == cur->data.size()) {
			std::unique_lock<std::mutex> lk(mu);
			cv_done.wait(lk, [&] { return (!order.empty() && order.front()->done) || (order.empty() && eof); });
			if (order.empty() || order.front()->bad) return false;
			cur = order.front();
			order.pop_front();
			cur_off = 0;
			cv_room.notify_one();
		}
		return true;
	}
	long bam_records(std::vector<BamRec> &out)
	{
		if (failed || !bgzf) return 0;
		if (!have_block()) return 0;                         // the end, or corrupt data: read() says which
		const size_t before = out.size();
		if (cur_off == 0 && cur->recs_end > 0) {                 // walked by the thread that inflated it
			out.insert(out.end(), cur->recs.begin(), cur->recs.end());
			cur_off = cur->recs_end;
		} else {
			cur_off = walk_records(cur->data.data(), cur->data.size(), cur_off, out);
		}
		return (long)(out.size() - before);
	}

	long read_parallel(uint8_t *dst, size_t n)
	{
		if (failed) return -1;
		size_t got = 0;
		while (got < n) {
			if (!cur || cur_off == cur->data.size()) {
				std::unique_lock<std::mutex> lk(mu);
				if (cur) { cur.reset(); }
				cv_done.wait(lk, [&] { return (!order.empty() && order.front()->done) || (order.empty() && eof); });
				if (order.empty()) { if (read_error) { failed = true; if (got == 0) return -1; } break; }
				cur = order.front();
				order.pop_front();
				cur_off = 0;
				cv_room.notify_one();
				if (cur->bad) { cur.reset(); failed = true; return got ? (long)got : -1; }
				continue;
			}
			const size_t take = std::min(n - got, cur->data.size() - cur_off);
			if (dst) memcpy(dst + got, cur->data.data() + cur_off, take);      // dst == nullptr: skip
			cur_off += take;
			got += take;
		}
		return (long)got;
	}

	long read_plain(uint8_t *dst, size_t n)
	{
		if (failed) return -1;
		size_t got = 0;
		while (got < n) {
			if (z.avail_in == 0 && !z_eof) {
				size_t r = 0;
				if (!read_full(fd, zin.data(), zin.size(), r)) { failed = true; return got ? (long)got : -1; }
				if (r == 0) z_eof = true;
				z.next_in = zin.data();
				z.avail_in = (uInt)r;
			}
			if (z.avail_in == 0 && z_eof) break;
			z.next_out = dst + got;
			z.avail_out = (uInt)std::min<size_t>(n - got, 1u << 30);
			const size_t before = z.avail_out;
			const int rc = ::inflate(&z, Z_NO_FLUSH);
			got += before - z.avail_out;
			if (rc == Z_STREAM_END) { inflateReset(&z); continue; }       // next gzip member
			if (rc != Z_OK && rc != Z_BUF_ERROR) { failed = true; return got ? (long)got : -1; }
			if (rc == Z_BUF_ERROR && z.avail_in == 0 && z_eof) break;
		}
		return (long)got;
	}
};

BgzfStream::BgzfStream(int fd, bool bam) : impl_(new Impl())
{
	Impl &m = *impl_;
	m.fd = fd;
	m.want_records = bam;
	m.head.clear();
	size_t xlen = 0;
	struct stat st;
	if (!getenv("SEQKIT_NO_MMAP") && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 28) {
		void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (mp != MAP_FAILED) {
			const uint8_t *h = static_cast<const uint8_t *>(mp);
			const size_t hl = (size_t)st.st_size < 12 + 65535 ? (size_t)st.st_size : 12 + 65535;
			if (Impl::bgzf_block_size(h, hl, xlen) != 0) { m.map = h; m.map_len = (size_t)st.st_size; (void)madvise(mp, m.map_len, MADV_SEQUENTIAL); }
			else munmap(mp, (size_t)st.st_size);             // gzip that is not BGZF, or no gzip at all: the descriptor is still at its start
		}
	}
	if (!m.map) Impl::read_header(fd, m.head);
	m.bgzf = m.map || Impl::bgzf_block_size(m.head.data(), m.head.size(), xlen) != 0;
	if (m.bgzf) {
		unsigned n = cpu_budget();
		if (const char *e = getenv("SEQKIT_THREADS")) n = (unsigned)atoi(e);
		if (n < 1) n = 1;
		if (n > 64) n = 64;
		m.max_in_flight = (size_t)n * 16;
		if (m.map) m.reader = std::thread([&m] { m.reader_main_mapped(); });
		else m.reader = std::thread([&m] { m.reader_main(); });
		for (unsigned i = 0; i < n; i++) m.workers.emplace_back([&m] { m.worker_main(); });
	} else {
		memset(&m.z, 0, sizeof m.z);
		inflateInit2(&m.z, 15 + 32);                         // gzip or zlib wrapper, detected; raw bytes fail like gzread's would not, see below
		m.zin.resize(std::max<size_t>(1 << 16, m.head.size()));
		memcpy(m.zin.data(), m.head.data(), m.head.size());
		m.z.next_in = m.zin.data();
		m.z.avail_in = (uInt)m.head.size();
	}
}

BgzfStream::~BgzfStream()
{
	Impl &m = *impl_;
	if (m.bgzf) {
		{
			std::lock_guard<std::mutex> lk(m.mu);
			m.stop = true;
			m.cv_work.notify_all(); m.cv_room.notify_all(); m.cv_done.notify_all();
		}
		if (m.reader.joinable()) m.reader.join();
		for (auto &t : m.workers) if (t.joinable()) t.join();
	} else {
		inflateEnd(&m.z);
	}
	if (m.map) munmap(const_cast<uint8_t *>(m.map), m.map_len);
	if (m.fd > 0) ::close(m.fd);
	delete impl_;
}

long BgzfStream::read(void *dst, size_t n)
{
	return impl_->bgzf ? impl_->read_parallel(static_cast<uint8_t *>(dst), n) : impl_->read_plain(static_cast<uint8_t *>(dst), n);
}

long BgzfStream::bam_records(std::vector<BamRec> &out)
{
	return impl_->bam_records(out);
}

long BgzfStream::skip(size_t n)
{
	if (impl_->bgzf) return impl_->read_parallel(nullptr, n);
	uint8_t buf[1 << 14];                                  // a plain zlib stream has to be inflated somewhere
	size_t got = 0;
	while (got < n) {
		const long r = impl_->read_plain(buf, std::min(n - got, sizeof buf));
		if (r < 0) return got ? (long)got : -1;
		if (r == 0) break;
		got += (size_t)r;
	}
	return (long)got;
}

// SEQKIT_GPU_DEFLATE=1: the writers' gzip members are deflated on the device (include/seqkit_hip.h: sk_bgzf_deflate).  The jobs of
// ALL files queue up as before; ONE thread takes what is there — hundreds of jobs: the kernel wants thousands of 64 KiB blocks, a
// wave each —, lays the jobs' bytes side by side, calls the library once and hands every job's members back to its file, which
// writes them in submission order as it always did.  A failure of the device path (no context, an error from the library) falls
// back to the CPU threads for the rest of the run, with one line on stderr: the output is the same bytes either way only after
// gunzip — which is what parity is defined on.
static std::atomic<bool> g_gz_on_device{false};
void gz_deflate_on_device(bool on) { g_gz_on_device.store(on); }
static bool gpu_deflate_wanted()
{
	if (const char *e = getenv("SEQKIT_GPU_DEFLATE")) return atoi(e) != 0;
	return g_gz_on_device.load();
}

void Pool::start()
{
	if (!threads.empty()) return;
	if (gpu_deflate_wanted()) {
		on_gpu = true;
		max_in_flight = 2048;                                  // jobs of 512 KiB: up to 1 GiB waiting for the device
		// a few batching threads, each with a context of its own: while one waits for the device, another lays its batch out and a
		// third writes its members to their files (one thread did all three in turn: 2.8 s for 8 M reads into 96 files, where the
		// CPU pool takes 2.6 — at 6 CPU-seconds instead of 28)
		int nt = 2;                                            // (round 6, after the writers' tails went in together: 2 threads 1.57 s, 3 1.65, 4 1.71, 6 2.03 for the same 8 M reads)
		if (const char *e = getenv("SEQKIT_GPU_DEFLATE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 8) nt = v; }
		for (int i = 0; i < nt; i++) threads.emplace_back([this] { run_gpu(); });
		return;
	}
	unsigned n = cpu_budget();
	if (const char *e = getenv("SEQKIT_THREADS")) n = (unsigned)atoi(e);
	if (n < 1) n = 1;
	if (n > 64) n = 64;
	max_in_flight = (size_t)n * 4;
	for (unsigned i = 0; i < n; i++) threads.emplace_back([this] { run(); });
}

void Pool::submit(Job &&j)
{
	std::unique_lock<std::mutex> lk(m);
	start();
	cv_room.wait(lk, [this] { return in_flight < max_in_flight; });     // bounds memory
	in_flight++;
	q.push_back(std::move(j));
	cv_job.notify_one();
}

void Pool::run()
{
	for (;;) {
		Job j;
		{
			std::unique_lock<std::mutex> lk(m);
			cv_job.wait(lk, [this] { return stop || !q.empty(); });
			if (q.empty()) return;
			j = std::move(q.front());
			q.pop_front();
		}
		// the copy that makes the member's bytes contiguous happens here, on a pool thread, not in the caller
		std::string joined;
		if (j.parts.size() > 1) {
			size_t total = 0;
			for (const std::string &p : j.parts) total += p.size();
			joined.reserve(total);
			for (const std::string &p : j.parts) joined += p;
		}
		static const std::string empty;
		std::string comp = gzip_member(j.parts.empty() ? empty : j.parts.size() == 1 ? j.parts[0] : joined);
		j.w->completed(j.seq, std::move(comp));
		{
			std::unique_lock<std::mutex> lk(m);
			in_flight--;
			cv_room.notify_one();
		}
	}
}

void Pool::run_gpu()
{
	sk_ctx *ctx = nullptr;
	bool broken = false;
	uint8_t *in = nullptr;                               // page-locked (sk_malloc_pinned): the batch's bytes go to the device by DMA from where they were laid out
	size_t in_cap = 0;
	std::vector<uint8_t> outbuf;
	std::vector<sk_deflate_block> blocks;
	std::vector<uint64_t> off;
	// SEQKIT_PROF=1: where this batching thread's time went
	const bool prof = getenv("SEQKIT_PROF") != nullptr;
	auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
	double t_idle = 0, t_ctx = 0, t_lay = 0, t_call = 0, t_hand = 0;
	size_t n_batches = 0, n_blocks = 0, n_bytes = 0;
	for (;;) {
		std::vector<Job> batch;
		const double tw0 = now();
		{
			std::unique_lock<std::mutex> lk(m);
			cv_job.wait(lk, [this] { return stop || !q.empty(); });
			if (q.empty()) break;
			// a little patience for a fuller batch: the kernel runs a wave per 64 KiB block, and a launch of a few blocks leaves the device idle
			if (q.size() < 64 && !stop) cv_job.wait_for(lk, std::chrono::milliseconds(2), [this] { return stop || q.size() >= 64; });
			while (!q.empty() && batch.size() < 256) { batch.push_back(std::move(q.front())); q.pop_front(); }
			if (!q.empty()) cv_job.notify_one();                          // (another batching thread takes the rest)
		}
		std::vector<std::string> comp(batch.size());
		bool done = false;
		const double tb0 = now();
		t_idle += tb0 - tw0;
		double tb1 = tb0, tb2 = tb0, tb3 = tb0;
		if (!broken) {
			if (!ctx) {
				const int dev = getenv("SEQKIT_GPU") ? atoi(getenv("SEQKIT_GPU")) : 0;
				if (sk_create(dev, &ctx) != SK_OK) { ctx = nullptr; broken = true; }
			}
			tb1 = tb2 = tb3 = now();
			if (ctx) {
				blocks.clear();
				size_t in_bytes = 0;
				for (const Job &jb : batch) for (const std::string &p : jb.parts) in_bytes += p.size();
				int rc = SK_OK;
				if (in_bytes + 8 > in_cap) {
					if (in) (void)sk_free_pinned(ctx, in);
					in = nullptr;
					in_cap = std::max<size_t>(in_bytes + in_bytes / 4 + 8, (size_t)32 << 20);
					void *pp = nullptr;
					rc = sk_malloc_pinned(ctx, in_cap, &pp);
					in = (uint8_t *)pp;
					if (rc != SK_OK) in_cap = 0;
				}
				std::vector<size_t> first(batch.size() + 1, 0);            // a job's first block
				size_t fill = 0;
				for (size_t j = 0; j < batch.size() && rc == SK_OK; j++) {
					first[j] = blocks.size();
					const size_t at = fill;
					size_t total = 0;
					for (const std::string &p : batch[j].parts) { memcpy(in + fill, p.data(), p.size()); fill += p.size(); total += p.size(); }
					for (size_t o = 0; o < total; o += kBgzfInput) blocks.push_back({(uint64_t)(at + o), (uint32_t)std::min(kBgzfInput, total - o), 0u});
				}
				first[batch.size()] = blocks.size();
				outbuf.resize(blocks.size() * (size_t)SK_DEFLATE_MAX_MEMBER + 64);
				off.assign(blocks.size() + 1, 0);
				tb2 = now();
				if (rc == SK_OK && !blocks.empty()) rc = sk_bgzf_deflate(ctx, in, fill, blocks.data(), (int64_t)blocks.size(), outbuf.data(), outbuf.size(), off.data());
				tb3 = now();
				n_batches++; n_blocks += blocks.size(); n_bytes += fill;
				if (rc == SK_OK) {
					for (size_t j = 0; j < batch.size(); j++)
						comp[j].assign(reinterpret_cast<const char *>(outbuf.data() + off[first[j]]), (size_t)(off[first[j + 1]] - off[first[j]]));
					done = true;
				} else {
					fprintf(stderr, "WARNING: the device's deflate failed (%s); the rest of the output is compressed on the CPU.\n", sk_last_error(ctx));
					broken = true;
				}
			}
		}
		for (size_t j = 0; j < batch.size(); j++) {
			if (!done) {                                                   // the CPU's members (the same decompressed bytes)
				std::string joined;
				for (const std::string &p : batch[j].parts) joined += p;
				comp[j] = gzip_member(joined);
			}
			batch[j].w->completed(batch[j].seq, std::move(comp[j]));
		}
		{
			std::unique_lock<std::mutex> lk(m);
			in_flight -= batch.size();
			cv_room.notify_all();
		}
		t_ctx += tb1 - tb0; t_lay += tb2 - tb1; t_call += tb3 - tb2; t_hand += now() - tb3;
	}
	const double td0 = now();
	if (ctx && in) (void)sk_free_pinned(ctx, in);
	if (ctx) sk_destroy(ctx);
	if (prof) fprintf(stderr, "deflate batcher: %zu batches, %zu blocks, %.0f MB; waiting for jobs %.3f s, context %.3f, laying the batch out %.3f, sk_bgzf_deflate %.3f, members to their files %.3f, sk_destroy %.3f\n",
	                  n_batches, n_blocks, n_bytes / 1e6, t_idle, t_ctx, t_lay, t_call, t_hand, now() - td0);
}

Pool::~Pool()
{
	{
		std::unique_lock<std::mutex> lk(m);
		stop = true;
		cv_job.notify_all();
	}
	for (auto &t : threads) t.join();
}

GzWriter::GzWriter(const std::string &path) : impl_(new Impl)
{
	impl_->fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
	if (impl_->fd < 0) { delete impl_; impl_ = nullptr; error("Cannot open file %s for writing.", path.c_str()); }
}

GzWriter::~GzWriter()
{
	close();
	delete impl_;
}

void GzWriter::Impl::submit()
{
	Job j{this, next_submit++, std::move(parts)};
	parts.clear();
	bytes = 0;
	open_tail = false;
	pool().submit(std::move(j));
}

void GzWriter::write(const char *p, size_t n)
{
	if (!impl_ || impl_->fd < 0) return;
	if (!impl_->open_tail) {
		impl_->parts.emplace_back();
		impl_->parts.back().reserve(kGzBlock + 4096);
		impl_->open_tail = true;
	}
	impl_->parts.back().append(p, n);
	impl_->bytes += n;
	if (impl_->bytes >= kGzBlock) impl_->submit();
}

// a whole string changes hands: no copy on the calling thread
void GzWriter::write(std::string &&s)
{
	if (!impl_ || impl_->fd < 0 || s.empty()) return;
	impl_->bytes += s.size();
	impl_->parts.push_back(std::move(s));
	impl_->open_tail = false;
	if (impl_->bytes >= kGzBlock) impl_->submit();
}

// what is buffered goes to the pool now; nothing is waited for (a command that closes many files hands in all their tails first: closing
// them one after the other made every tail a launch of its own on the device path — 96 files, 7 ms each)
void GzWriter::flush_tail()
{
	if (!impl_ || impl_->fd < 0) return;
	if (impl_->bytes > 0) impl_->submit();
}

void GzWriter::close()
{
	if (!impl_ || impl_->fd < 0) return;
	if (impl_->bytes > 0) impl_->submit();
	{
		std::unique_lock<std::mutex> lk(impl_->m);
		impl_->cv.wait(lk, [this] { return impl_->next_write == impl_->next_submit; });
	}
	// BGZF's end-of-file marker: an empty block (also what an empty file consists of: one empty member, like
	// `gzip -c < /dev/null`)
	for (size_t off = 0; off < sizeof kBgzfEof;) {
		const ssize_t w = ::write(impl_->fd, kBgzfEof + off, sizeof kBgzfEof - off);
		if (w <= 0) break;
		off += (size_t)w;
	}
	::close(impl_->fd);
	impl_->fd = -1;
}

// ---- stdout ------------------------------------------------------------------------------------------------
void Out::write(const char *p, size_t n)
{
	if (n >= (1u << 20)) {                                     // a block's worth: no second copy through the buffer
		flush();
		size_t off = 0;
		while (off < n) {
			const ssize_t w = ::write(1, p + off, n - off);
			if (w <= 0) break;
			off += (size_t)w;
		}
		return;
	}
	buf_.append(p, n);
	if (buf_.size() >= (1u << 20)) flush();
}
void Out::flush()
{
	size_t off = 0;
	while (off < buf_.size()) {
		const ssize_t w = ::write(1, buf_.data() + off, buf_.size() - off);
		if (w <= 0) break;
		off += (size_t)w;
	}
	buf_.clear();
}
Out &out()
{
	static Out o;
	return o;
}

// ---- GPU ---------------------------------------------------------------------------------------------------
// The contexts take a few hundred ms to create (runtime start-up, code object load).  gpu_warmup() starts that in the
// background as soon as a command knows it will need the device, so that it overlaps opening and reading the input;
// a failure stays silent until a context is actually asked for.
struct GpuInit { int rc = SK_OK; std::vector<sk_ctx *> ctxs; std::string err; };
static GpuInit create_ctxs()
{
	GpuInit r;
	std::vector<int> devs;
	if (const char *e = getenv("SEQKIT_GPUS")) {
		for (const char *p = e; *p;) {
			char *end = nullptr;
			const long d = strtol(p, &end, 10);
			if (end == p) break;
			devs.push_back((int)d);
			p = (*end == ',') ? end + 1 : end;
			if (*end && *end != ',') break;
		}
	}
	if (devs.empty()) devs.push_back(getenv("SEQKIT_GPU") ? atoi(getenv("SEQKIT_GPU")) : 0);
	int per = getenv("SEQKIT_CTXS_PER_GPU") ? atoi(getenv("SEQKIT_CTXS_PER_GPU")) : 2;
	if (per < 1) per = 1;
	if (per > 8) per = 8;
	for (int k = 0; k < per && r.rc == SK_OK; k++)          // slot order d0 d1 ... d0 d1 ...: consecutive blocks go to different devices
		for (int d : devs) {
			sk_ctx *c = nullptr;
			r.rc = sk_create(d, &c);
			if (r.rc != SK_OK) { r.err = sk_last_error(nullptr); break; }
			r.ctxs.push_back(c);
		}
	return r;
}
static std::mutex g_warm_m;
static std::shared_future<GpuInit> g_warm;

static const double g_process_start = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }();
double process_start_s() { return g_process_start; }
static const struct ExitProf {                    // SEQKIT_PROF: when the process is through with its static destructors
	~ExitProf()
	{
		if (!getenv("SEQKIT_PROF")) return;
		timespec ts;
		clock_gettime(CLOCK_MONOTONIC, &ts);
		fprintf(stderr, "process: %.3f s from static initialisation to the last static destructor of host_common\n", ts.tv_sec + ts.tv_nsec * 1e-9 - g_process_start);
	}
} g_exit_prof;

void gpu_warmup()
{
	std::lock_guard<std::mutex> lk(g_warm_m);
	if (!g_warm.valid()) g_warm = std::async(std::launch::async, create_ctxs).share();
}

namespace {
struct GpuPool {
	std::vector<sk_ctx *> ctxs;
	std::vector<char> busy;
	size_t next = 0;
	std::mutex m;
	std::condition_variable cv;
};
GpuPool &gpu_pool()
{
	static GpuPool p;
	static std::once_flag once;
	std::call_once(once, [] {
		gpu_warmup();
		const GpuInit r = g_warm.get();
		if (r.rc != SK_OK) error("No usable MI355X for the seqkit HIP path (%s); this build has no CPU fallback.", r.err.c_str());
		p.ctxs = r.ctxs;
		p.busy.assign(p.ctxs.size(), 0);
	});
	return p;
}
}  // namespace

sk_ctx *gpu() { return gpu_pool().ctxs[0]; }
size_t gpu_slots() { return gpu_pool().ctxs.size(); }
sk_ctx *gpu_slot(size_t i) { return gpu_pool().ctxs[i]; }
void gpu_for_each(const std::function<void(sk_ctx *)> &fn)
{
	for (sk_ctx *c : gpu_pool().ctxs) fn(c);
}

GpuLease::GpuLease()
{
	GpuPool &p = gpu_pool();
	std::unique_lock<std::mutex> lk(p.m);
	for (;;) {
		for (size_t k = 0; k < p.ctxs.size(); k++) {
			const size_t s = (p.next + k) % p.ctxs.size();
			if (!p.busy[s]) {
				p.busy[s] = 1;
				p.next = s + 1;
				slot_ = s;
				ctx_ = p.ctxs[s];
				return;
			}
		}
		p.cv.wait(lk);
	}
}

GpuLease::GpuLease(size_t slot)
{
	GpuPool &p = gpu_pool();
	std::unique_lock<std::mutex> lk(p.m);
	p.cv.wait(lk, [&] { return !p.busy[slot]; });
	p.busy[slot] = 1;
	slot_ = slot;
	ctx_ = p.ctxs[slot];
}

GpuLease::~GpuLease()
{
	GpuPool &p = gpu_pool();
	{
		std::lock_guard<std::mutex> lk(p.m);
		p.busy[slot_] = 0;
	}
	p.cv.notify_all();
}

// ---- pinned staging ---------------------------------------------------------------------------------------------
struct PinnedArena::Buf { uint8_t *p = nullptr; size_t cap = 0, want = 0; };
namespace {
std::mutex g_arena_m;
std::vector<PinnedArena::Buf *> g_arena_free;
}  // namespace

PinnedArena::PinnedArena(size_t hint_bytes)
{
	{
		std::lock_guard<std::mutex> lk(g_arena_m);
		if (!g_arena_free.empty()) { buf_ = g_arena_free.back(); g_arena_free.pop_back(); }
		else buf_ = new Buf();
	}
	if (hint_bytes > buf_->want) buf_->want = hint_bytes;
	if (buf_->want > buf_->cap) {                             // the last use overflowed: grow now, while nothing points into it
		if (buf_->p) (void)sk_free_pinned(gpu(), buf_->p);
		buf_->p = nullptr;
		void *q = nullptr;
		const size_t cap = buf_->want + (buf_->want >> 2);
		if (sk_malloc_pinned(gpu(), cap, &q) != SK_OK) error("Cannot allocate %zu bytes of pinned host memory (%s).", cap, sk_last_error(gpu()));
		buf_->p = static_cast<uint8_t *>(q);
		buf_->cap = cap;
	}
}

PinnedArena::~PinnedArena()
{
	for (void *q : extra_) (void)sk_free_pinned(gpu(), q);
	if (used_ > buf_->want) buf_->want = used_;
	std::lock_guard<std::mutex> lk(g_arena_m);
	g_arena_free.push_back(buf_);
}

uint8_t *PinnedArena::take(size_t bytes)
{
	const size_t at = (used_ + 63) & ~(size_t)63;
	used_ = at + bytes;
	if (used_ <= buf_->cap) return buf_->p + at;
	void *q = nullptr;                                        // does not fit: its own block for this use, a bigger main buffer next time
	if (sk_malloc_pinned(gpu(), bytes ? bytes : 1, &q) != SK_OK) error("Cannot allocate %zu bytes of pinned host memory (%s).", bytes, sk_last_error(gpu()));
	extra_.push_back(q);
	return static_cast<uint8_t *>(q);
}

size_t trim_end_len(const char *p, size_t n)
{
	// fast path: ASCII tail
	while (n > 0) {
		const uint8_t b = (uint8_t)p[n - 1];
		if (b >= 0x80) return trim_end_len(std::string(p, n));
		if (!((b >= 0x09 && b <= 0x0D) || b == 0x20)) break;
		n--;
	}
	return n;
}

bool is_ascii(const char *p, size_t n)
{
	for (size_t i = 0; i < n; i++)
		if ((uint8_t)p[i] >= 0x80) return false;
	return true;
}

// ---- block pipeline ------------------------------------------------------------------------------------------
namespace {

// raw byte source with the FileReader path rules (src/common.rs:88-103)
struct RawSource {
	gzFile gz = nullptr;
	std::unique_ptr<BgzfStream> bz;                           // a *.gz input that is BGZF: inflated block-parallel
	int fd = -1;
	explicit RawSource(const std::string &path)
	{
		if (path == "-") { fd = 0; return; }
		fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) error("Cannot open file %s for reading.", path.c_str());
		if (path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
			if (fd_is_bgzf(fd)) {
				bz.reset(new BgzfStream(fd));             // takes the descriptor
				fd = -1;
				return;
			}
			gz = gzdopen(fd, "rb");
			if (!gz) error("Cannot start gunzip process.");
			gzbuffer(gz, 1 << 20);
		}
	}
	~RawSource() { if (gz) gzclose(gz); else if (fd > 0) close(fd); }
	bool compressed() const { return gz != nullptr || bz != nullptr; }
	size_t read_some(char *dst, size_t n)
	{
		long r;
		if (bz) r = bz->read(dst, n);
		else if (gz) r = gzread(gz, dst, (unsigned)std::min<size_t>(n, 1u << 30));
		else do { r = (long)read(fd, dst, n); } while (r < 0 && errno == EINTR);
		if (r < 0) error("I/O error while reading from file.");
		return (size_t)r;
	}
};

}  // namespace

static size_t count_newlines_swar(const char *p, size_t n)
{
	size_t cnt = 0, i = 0;
	const uint64_t k7f = 0x7F7F7F7F7F7F7F7Full, knl = 0x0A0A0A0A0A0A0A0Aull;
	for (; i + 8 <= n; i += 8) {
		uint64_t w;
		memcpy(&w, p + i, 8);
		const uint64_t x = w ^ knl;                              // zero bytes where the input has a newline
		const uint64_t t = ~(((x & k7f) + k7f) | x | k7f);      // 0x80 exactly in the zero bytes
		cnt += (size_t)__builtin_popcountll(t);
	}
	for (; i < n; i++) cnt += p[i] == '\n';
	return cnt;
}

// 128 bytes per iteration where the CPU has AVX2 (the cutter thread of the block pipeline looks at every input byte
// once, here: 2.9 -> 9 GB/s)
__attribute__((target("avx2"))) static size_t count_newlines_avx2(const char *p, size_t n)
{
	size_t cnt = 0, i = 0;
	const __m256i nl = _mm256_set1_epi8('\n');
	for (; i + 128 <= n; i += 128) {
		const __m256i a = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), nl);
		const __m256i b = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 32)), nl);
		const __m256i c = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 64)), nl);
		const __m256i d = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 96)), nl);
		cnt += (size_t)__builtin_popcountll((uint64_t)(uint32_t)_mm256_movemask_epi8(a) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(b) << 32));
		cnt += (size_t)__builtin_popcountll((uint64_t)(uint32_t)_mm256_movemask_epi8(c) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(d) << 32));
	}
	return cnt + count_newlines_swar(p + i, n - i);
}

static size_t count_newlines(const char *p, size_t n)
{
	static const bool avx2 = __builtin_cpu_supports("avx2");
	return avx2 ? count_newlines_avx2(p, n) : count_newlines_swar(p, n);
}

// consume up to `want` lines of [p, p+n): returns the bytes consumed (ending right after a newline, or n when the
// buffer runs out first) and subtracts the lines found from `want`
static size_t take_lines_swar(const char *p, size_t n, size_t &want)
{
	size_t i = 0;
	const uint64_t k7f = 0x7F7F7F7F7F7F7F7Full, knl = 0x0A0A0A0A0A0A0A0Aull;
	while (want > 0 && i + 8 <= n) {
		uint64_t w;
		memcpy(&w, p + i, 8);
		const uint64_t x = w ^ knl;
		const uint64_t t = ~(((x & k7f) + k7f) | x | k7f);
		const size_t c = (size_t)__builtin_popcountll(t);
		if (c < want) { want -= c; i += 8; continue; }
		break;                                                   // the last wanted newline is inside this word
	}
	while (want > 0 && i < n) {
		if (p[i] == '\n') want--;
		i++;
	}
	return i;
}

__attribute__((target("avx2"))) static size_t take_lines_avx2(const char *p, size_t n, size_t &want)
{
	size_t i = 0;
	const __m256i nl = _mm256_set1_epi8('\n');
	while (want > 0 && i + 64 <= n) {
		const __m256i a = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), nl);
		const __m256i b = _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 32)), nl);
		const size_t c = (size_t)__builtin_popcountll((uint64_t)(uint32_t)_mm256_movemask_epi8(a) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(b) << 32));
		if (c < want) { want -= c; i += 64; continue; }
		break;                                                   // the last wanted newline is inside these 64 bytes
	}
	return i + take_lines_swar(p + i, n - i, want);
}

static size_t take_lines(const char *p, size_t n, size_t &want)
{
	static const bool avx2 = __builtin_cpu_supports("avx2");
	return avx2 ? take_lines_avx2(p, n, want) : take_lines_swar(p, n, want);
}

// ---- Bytes: recycled block buffers ----
namespace {
struct FreeBufs {
	std::mutex m;
	std::vector<std::pair<char *, size_t>> list;
	~FreeBufs() { for (auto &e : list) free(e.first); }
};
FreeBufs &free_bufs()
{
	static FreeBufs f;
	return f;
}
constexpr size_t kMaxFreeBufs = 96;
}  // namespace

void Bytes::borrow(const char *p, size_t n)
{
	this->~Bytes();
	p_ = const_cast<char *>(p);
	n_ = n;
	cap_ = 0;                                                // nothing may be appended to a view
	owned_ = false;
}

Bytes::~Bytes()
{
	if (!p_ || !owned_) { p_ = nullptr; n_ = cap_ = 0; owned_ = true; return; }
	FreeBufs &f = free_bufs();
	std::lock_guard<std::mutex> lk(f.m);
	if (f.list.size() < kMaxFreeBufs) f.list.emplace_back(p_, cap_);
	else free(p_);
	p_ = nullptr;
	n_ = cap_ = 0;
}

Bytes &Bytes::operator=(Bytes &&o) noexcept
{
	if (this != &o) {
		this->~Bytes();
		p_ = o.p_; n_ = o.n_; cap_ = o.cap_; owned_ = o.owned_;
		o.p_ = nullptr; o.n_ = o.cap_ = 0; o.owned_ = true;
	}
	return *this;
}

void Bytes::reserve(size_t cap)
{
	if (!owned_) { p_ = nullptr; n_ = cap_ = 0; owned_ = true; }      // a view cannot grow: start an own buffer (callers clear() first)
	if (cap <= cap_) return;
	if (n_ == 0) {                                           // nothing to keep: a recycled buffer that is big enough will do
		FreeBufs &f = free_bufs();
		std::lock_guard<std::mutex> lk(f.m);
		size_t best = f.list.size();
		for (size_t i = 0; i < f.list.size(); i++)
			if (f.list[i].second >= cap && (best == f.list.size() || f.list[i].second < f.list[best].second)) best = i;
		if (best < f.list.size()) {
			if (p_) f.list.emplace_back(p_, cap_);             // (cannot overflow the list: one was just taken... almost)
			p_ = f.list[best].first;
			cap_ = f.list[best].second;
			f.list.erase(f.list.begin() + (long)best);
			return;
		}
	}
	size_t nc = cap_ * 2 > cap ? cap_ * 2 : cap;
	char *np = static_cast<char *>(realloc(p_, nc));
	if (!np) error("out of memory");
	p_ = np;
	cap_ = nc;
}

struct RecordBlocks::Impl {
	RawSource src;
	int lpr;
	Bytes carry;                      // bytes read beyond the last block handed out
	size_t biggest = 0;               // the largest block so far: the next one reserves that much at once
	bool eof = false;
	const char *map = nullptr;        // a regular file is mapped: blocks are views of the mapping, nothing is copied
	size_t map_n = 0, map_off = 0;
	Impl(const std::string &path, int l) : src(path), lpr(l)
	{
		struct stat st;
		if (!src.compressed() && src.fd > 0 && !getenv("SEQKIT_NO_MMAP") && fstat(src.fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
			void *a = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, src.fd, 0);
			if (a != MAP_FAILED) { map = static_cast<const char *>(a); map_n = (size_t)st.st_size; (void)madvise(a, map_n, MADV_SEQUENTIAL); }
		}
	}
	~Impl() { if (map) munmap(const_cast<char *>(map), map_n); }
};

RecordBlocks::RecordBlocks(const std::string &path, int lines_per_record) : impl_(new Impl(path, lines_per_record)) {}
RecordBlocks::~RecordBlocks() { delete impl_; }

bool RecordBlocks::next(size_t nrec, Bytes &blk)
{
	// The block is put together in the caller's buffer: what the previous call left over, then fresh chunks until it
	// holds the wanted lines; only the tail beyond the cut (less than one chunk) is copied, into the next call's start.
	Impl &m = *impl_;
	if (m.map) {
		blk.clear();
		if (m.map_off == m.map_n) return false;
		size_t left = nrec * (size_t)m.lpr;
		const size_t cut = take_lines(m.map + m.map_off, m.map_n - m.map_off, left);      // the wanted lines, or everything that is left
		blk.borrow(m.map + m.map_off, cut);
		m.map_off += cut;
		return cut > 0;
	}
	const size_t chunk = 4u << 20;
	blk.clear();
	blk.reserve(std::max(m.biggest, m.carry.size()) + chunk);
	blk.append(m.carry.data(), m.carry.size());
	m.carry.clear();
	const size_t want_lines = nrec * (size_t)m.lpr;
	size_t scan = 0, lines = 0, cut;
	for (;;) {
		if (scan < blk.size() && lines < want_lines) {
			size_t left = want_lines - lines;
			scan += take_lines(blk.data() + scan, blk.size() - scan, left);
			lines = want_lines - left;
		}
		if (lines >= want_lines) { cut = scan; break; }
		if (m.eof) { cut = blk.size(); break; }
		const size_t old = blk.size();
		blk.reserve(old + chunk);
		const size_t r = m.src.read_some(blk.data() + old, chunk);
		blk.set_size(old + r);
		if (r == 0) m.eof = true;
	}
	if (cut == 0) { blk.clear(); return false; }
	m.carry.append(blk.data() + cut, blk.size() - cut);
	blk.set_size(cut);
	if (cut > m.biggest) m.biggest = cut;
	return true;
}

void run_block_pipeline(const std::string &path, int lines_per_record, const BlockFn &fn)
{
	RawSource src(path);
	size_t block_bytes = 8u << 20;
	unsigned nthreads = cpu_budget();
	if (const char *e = getenv("SEQKIT_THREADS")) nthreads = (unsigned)atoi(e);
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 32) nthreads = 32;
	if (const char *e = getenv("SEQKIT_BLOCK_BYTES")) block_bytes = (size_t)atoll(e);      // tests use tiny blocks

	// Three roles.  This thread reads and cuts (it looks at every input byte once, to count newlines); workers turn
	// blocks into output; a writer thread takes the results in input order and writes them, so that reading the input and
	// writing the output overlap.  Blocks are plain uninitialised buffers: value-initialising 8 MiB per block and copying
	// every result through the output buffer were half of this thread's time.
	struct Block { std::unique_ptr<char[]> p; const char *d = nullptr; size_t n = 0, cap = 0; };      // d: the bytes (p's, or a piece of the mapped file)
	struct Pending { std::shared_ptr<Block> data; std::future<std::shared_ptr<BlockResult>> fut; };
	std::deque<Pending> inflight;
	std::mutex m;
	std::condition_variable cv_push, cv_pop;
	bool closed = false, failed = false;
	std::shared_ptr<BlockResult> failure;
	std::thread writer([&]() {
		for (;;) {
			Pending pd;
			{
				std::unique_lock<std::mutex> lk(m);
				cv_push.wait(lk, [&] { return !inflight.empty() || closed; });
				if (inflight.empty()) return;
				pd = std::move(inflight.front());
				inflight.pop_front();
			}
			cv_pop.notify_one();
			std::shared_ptr<BlockResult> r = pd.fut.get();
			out().write(r->out);
			if (!r->err.empty()) {
				// the output of the records before the bad one is out; let the blocks in flight finish, then stop
				std::unique_lock<std::mutex> lk(m);
				failed = true;
				failure = r;
				while (!inflight.empty()) {
					Pending q = std::move(inflight.front());
					inflight.pop_front();
					lk.unlock();
					q.fut.wait();
					lk.lock();
				}
				lk.unlock();
				cv_pop.notify_all();
				return;
			}
		}
	});

	// SEQKIT_PROF=1: where this thread's time went, on stderr when the input is through
	double t_alloc = 0, t_read = 0, t_count = 0, t_wait = 0, t_spawn = 0;
	auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
	const bool prof = getenv("SEQKIT_PROF") != nullptr;
	auto stopped = [&] { std::lock_guard<std::mutex> lk(m); return failed; };
	auto hand_over = [&](const std::shared_ptr<Block> &blk, bool last) {
		double t0 = now();
		Pending pd;
		pd.data = blk;
		pd.fut = std::async(std::launch::async, [blk, last, &fn]() {
			auto res = std::make_shared<BlockResult>();
			fn(blk->d, blk->n, last, *res);
			return res;
		});
		t_spawn += now() - t0; t0 = now();
		{
			std::unique_lock<std::mutex> lk(m);
			cv_pop.wait(lk, [&] { return inflight.size() < nthreads || failed; });
			inflight.push_back(std::move(pd));         // (after a failure the writer is gone: the destructor of the future waits)
		}
		cv_push.notify_one();
		t_wait += now() - t0;
	};
	// where a block of whole records ends: count its newlines, then walk back from the end over the lines of the last,
	// incomplete record (0: not even one whole record in it)
	auto whole_records = [&](const char *base, size_t n) -> size_t {
		const size_t lines = count_newlines(base, n);
		size_t drop = lines % (size_t)lines_per_record;
		if (lines <= drop) return 0;
		const char *nl = static_cast<const char *>(memrchr(base, '\n', n));
		while (drop > 0) { nl = static_cast<const char *>(memrchr(base, '\n', (size_t)(nl - base))); drop--; }
		return (size_t)(nl - base) + 1;
	};

	// A regular file is mapped: the blocks are pieces of the mapping, nothing is copied on this thread (read(2) was half
	// of its time), and the workers take the page faults of their own blocks.  Pipes and gzip streams are read.
	struct Mapping { const char *p = nullptr; size_t n = 0; ~Mapping() { if (p) munmap(const_cast<char *>(p), n); } } map;
	if (!src.compressed() && src.fd > 0 && !getenv("SEQKIT_NO_MMAP")) {
		struct stat st;
		if (fstat(src.fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
			void *a = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, src.fd, 0);
			if (a != MAP_FAILED) {
				map.p = static_cast<const char *>(a);
				map.n = (size_t)st.st_size;
				(void)madvise(a, map.n, MADV_SEQUENTIAL);
			}
		}
	}
	if (map.p) {
		size_t off = 0;
		while (off < map.n && !stopped()) {
			double t0 = now();
			const size_t want = std::min(block_bytes, map.n - off);
			const bool last = off + want == map.n;
			size_t cut = want;
			if (!last) {
				cut = whole_records(map.p + off, want);
				if (cut == 0) { block_bytes *= 2; continue; }       // a record larger than the block: look further
			}
			t_count += now() - t0;
			auto blk = std::make_shared<Block>();
			blk->d = map.p + off;
			blk->n = cut;
			hand_over(blk, last);
			off += cut;
		}
	}
	std::string carry;
	bool eof = map.p != nullptr;
	while (!eof) {
		if (stopped()) break;
		double t0 = now();
		// fill one block: previous carry + fresh bytes, then cut after the last newline that completes a record
		auto blk = std::make_shared<Block>();
		blk->cap = carry.size() + block_bytes;
		blk->p.reset(new char[blk->cap]);
		memcpy(blk->p.get(), carry.data(), carry.size());
		const size_t start = carry.size();
		size_t got = 0;
		t_alloc += now() - t0; t0 = now();
		while (got < block_bytes) {
			const size_t r = src.read_some(blk->p.get() + start + got, block_bytes - got);
			if (r == 0) { eof = true; break; }
			got += r;
		}
		blk->n = start + got;
		carry.clear();
		t_read += now() - t0; t0 = now();
		if (!eof) {
			const char *base = blk->p.get();
			const size_t cut = whole_records(base, blk->n);      // keep whole records only
			if (cut == 0) {                    // a record larger than the block: grow and retry
				carry.assign(base, blk->n);
				block_bytes *= 2;
				continue;
			}
			carry.assign(base + cut, blk->n - cut);
			blk->n = cut;
		}
		if (blk->n == 0 && eof) break;
		t_count += now() - t0;
		blk->d = blk->p.get();
		hand_over(blk, eof);
	}
	{
		std::lock_guard<std::mutex> lk(m);
		closed = true;
	}
	cv_push.notify_one();
	writer.join();
	if (prof) fprintf(stderr, "block pipeline, reading thread: allocate %.3f read %.3f cut %.3f hand over %.3f wait for a free worker %.3f s\n", t_alloc, t_read, t_count, t_spawn, t_wait);
	{
		std::unique_lock<std::mutex> lk(m);
		while (!inflight.empty()) {                    // blocks queued after the writer stopped
			Pending q = std::move(inflight.front());
			inflight.pop_front();
			lk.unlock();
			q.fut.wait();
			lk.lock();
		}
	}
	if (failure) {
		if (failure->err_code == 101) panic(failure->err.c_str());
		error("%s", failure->err.c_str());
	}
}

}  // namespace host
