// sam — the `sam` binary of the reference for `sam statistics` and `sam fragment lengths`, with the per-record
// flag / TLEN reduction done by the MI355X library behind include/seqkit_hip.h (dispatch: src/sam_main.rs:50-53).
//
//   sam statistics [--on-target=BED] <bam_file>                         src/sam_statistics.rs:14-116
//   sam fragment lengths [--max-frag-size=F] [--reads=N] <bam_file>     src/sam_fragment_lengths.rs:14-48
//   sam fragments [--min-size=N] [--max-size=N] <bam_file>              src/sam_fragments.rs:14-43   (§8f f2)
//   sam count [--min-mapq=N] [--max-frag-len=N] [--single-end] [--center] <bam_file> <regions.bed>   src/sam_count.rs:20-130 (§8f f2)
//   sam to [interleaved] raw|fasta|fastq <bam_file> [<out_prefix>]      src/sam_to_fastq.rs:61-149   (§8f f4)
//
// The reference reads BAM through rust-htslib; this host walks the BGZF/BAM container itself (SAMv1 §4.2: BGZF is a
// series of gzip members; after the header every record is block_size:u32 + a 32-byte fixed core) and hands the core
// columns flag / refID / next_refID / tlen to the device as SoA batches.  Order-dependent pieces stay here, as
// SURVEY.md §8(e) lists them: the --reads=N early stop and the --on-target sweep (S2, not part of the device path).
#include <unistd.h>
#include <malloc.h>
#include <fcntl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_common.h"

using host::error;
using host::panic;

static const char *USAGE_TOP =
	"\nUsage:\n"
	"  sam merge <bam_files>...\n  sam consensus <bam_file>\n  sam count <bam_file> <regions.bed>\n  sam coverage histogram <bam_file>\n"
	"  sam fragments <bam_file>\n  sam fragment lengths <bam_file>\n  sam mark duplicates <bam_file>\n  sam minimize <bam_file>\n"
	"  sam statistics <bam_file>\n  sam subsample <bam_file> <fraction>  \n  sam tags from qname <bam_file>\n  sam qname from tags <bam_file>\n"
	"  sam trim qnames <bam_file>  \n\nExtract reads from BAM files:  \n  sam to fasta <bam_file> <out_prefix>\n  sam to fastq <bam_file> <out_prefix>  \n"
	"  sam to interleaved fasta <bam_file>\n  sam to interleaved fastq <bam_file>\n  sam to interleaved raw <bam_file>\n  sam to raw <bam_file> <out_prefix>\n";
static const char *USAGE_STATS =
	"\nUsage:\n  sam statistics [options] <bam_file>\n\nOptions:\n  --on-target=BED   Count on-target% for regions in BED file [optional]\n";
static const char *USAGE_FRAG =
	"\nUsage:\n  sam fragment lengths [options] <bam_file>\n\nOptions:\n"
	"  --max-frag-size=F     Maximum fragment size [default: 5000]\n"
	"  --reads=N             Finish after analyzing this many reads [default: Inf]\n";

static void check(int rc, const char *what)
{
	if (rc != SK_OK) error("%s failed: %s", what, sk_last_error(host::gpu()));
}

// ---- BGZF/BAM container walk ---------------------------------------------------------------------------------
struct BamCore { int32_t tid, pos; uint16_t flag; int32_t mtid, mpos, tlen, end_pos; uint8_t mapq; };

class BamStream {
public:
	explicit BamStream(const std::string &path) : path_(path)
	{
		int fd = 0;
		if (path != "-") fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) error("Cannot open BAM file '%s'", path.c_str());
		bz_.reset(new host::BgzfStream(fd, true));
		uint8_t h[8];
		if (!get(h, 8) || memcmp(h, "BAM\1", 4) != 0) open_fail();
		if (!skip(le32(h + 4))) open_fail();
		if (!get(h, 4)) open_fail();
		const uint32_t n_ref = le32(h);
		for (uint32_t i = 0; i < n_ref; i++) {
			if (!get(h, 4)) open_fail();
			const uint32_t l_name = le32(h);
			if (l_name > (1u << 20)) open_fail();               // a reference name of megabytes is a damaged header, not an allocation to attempt
			std::string name(l_name, '\0');
			if (l_name && !get(reinterpret_cast<uint8_t *>(&name[0]), l_name)) open_fail();
			if (!name.empty() && name.back() == '\0') name.pop_back();
			names.push_back(name);
			if (!get(h, 4)) open_fail();
		}
	}
	// Record errors ("BAM file ended prematurely." / "Invalid BAM record.", src/common.rs:150-154) end the stream: next()
	// returns false and the caller, once it has written what the records before the error produce, calls
	// raise_deferred() — the point the record-at-a-time reference would have reached.
	void raise_deferred() const { if (!err_.empty()) error("%s", err_.c_str()); }
	// next record; want_end computes cigar().end_pos() (needed only by the on-target sweep for unpaired reads)
	bool next(BamCore &c, bool want_end)
	{
		if (!err_.empty()) return false;
		// block_size and the 32-byte core in one read; a short one is sorted out by the rules of the two reads it replaces
		uint8_t hc[36];
		const long r = bz_->read(hc, 36);
		if (r == 0) return false;
		if (r < 0) return rd_fail("Invalid BAM record.");
		if (r < 4) return rd_fail("BAM file ended prematurely.");
		const uint32_t block_size = le32(hc);
		if (block_size < 32) return rd_fail("Invalid BAM record.");
		if (r < 36 && !need(hc + r, (size_t)(36 - r))) return false;
		const uint8_t *core = hc + 4;
		c.tid = (int32_t)le32(core + 0);
		c.pos = (int32_t)le32(core + 4);
		const uint32_t l_read_name = core[8];
		c.mapq = core[9];
		const uint32_t n_cigar = (uint32_t)core[12] | ((uint32_t)core[13] << 8);
		c.flag = (uint16_t)(core[14] | (core[15] << 8));
		c.mtid = (int32_t)le32(core + 20);
		c.mpos = (int32_t)le32(core + 24);
		c.tlen = (int32_t)le32(core + 28);
		c.end_pos = c.pos;
		uint32_t rest = block_size - 32;
		if (want_end && rest >= l_read_name + 4 * n_cigar) {
			var_.resize(l_read_name + 4 * n_cigar);
			if (!need(var_.data(), var_.size())) return false;
			rest -= (uint32_t)var_.size();
			int64_t e = c.pos;
			for (uint32_t k = 0; k < n_cigar; k++) {
				const uint32_t op = le32(var_.data() + l_read_name + 4 * k);
				const uint32_t code = op & 15, len = op >> 4;
				if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) e += len;   // M D N = X consume the reference
			}
			c.end_pos = (int32_t)e;
		}
		return skip(rest);
	}
	// The next records, one call for many: every record that lies wholly inside the current inflated block (walked by the
	// thread that inflated it, host::BgzfStream::bam_records), or — a record that straddles blocks, the end of the data, an
	// invalid record, or want_end — one record by next().  false as next(): the stream has ended or failed.
	bool next_chunk(std::vector<BamCore> &out, bool want_end)
	{
		out.clear();
		if (!err_.empty()) return false;
		if (!want_end) {
			recs_.clear();
			if (bz_->bam_records(recs_) > 0) {
				out.resize(recs_.size());
				for (size_t i = 0; i < recs_.size(); i++) {
					const host::BgzfStream::BamRec &r = recs_[i];
					BamCore &c = out[i];
					c.tid = r.tid; c.pos = r.pos; c.flag = r.flag; c.mtid = r.mtid; c.mpos = r.mpos; c.tlen = r.tlen; c.end_pos = r.pos; c.mapq = r.mapq;
				}
				return true;
			}
		}
		BamCore c;
		if (!next(c, want_end)) return false;
		out.push_back(c);
		return true;
	}
	// next record with its variable part (qname, cigar, packed bases, qualities, aux) in `body`
	struct Var { uint32_t l_read_name, n_cigar, l_seq; };
	bool next_full(BamCore &c, Var &v, std::vector<uint8_t> &body)
	{
		if (!err_.empty()) return false;
		uint8_t hc[36];                                          // block_size and the core in one read, as in next()
		const long r = bz_->read(hc, 36);
		if (r == 0) return false;
		if (r < 0) return rd_fail("Invalid BAM record.");
		if (r < 4) return rd_fail("BAM file ended prematurely.");
		const uint32_t block_size = le32(hc);
		if (block_size < 32) return rd_fail("Invalid BAM record.");
		if (r < 36 && !need(hc + r, (size_t)(36 - r))) return false;
		const uint8_t *core = hc + 4;
		c.tid = (int32_t)le32(core + 0);
		c.pos = (int32_t)le32(core + 4);
		v.l_read_name = core[8];
		c.mapq = core[9];
		v.n_cigar = (uint32_t)core[12] | ((uint32_t)core[13] << 8);
		c.flag = (uint16_t)(core[14] | (core[15] << 8));
		v.l_seq = le32(core + 16);
		c.mtid = (int32_t)le32(core + 20);
		c.mpos = (int32_t)le32(core + 24);
		c.tlen = (int32_t)le32(core + 28);
		c.end_pos = c.pos;
		const uint32_t rest = block_size - 32;
		// htslib bam_read1: a record whose variable part cannot hold its own fields is invalid
		if (v.l_read_name < 1 || v.l_seq > 0x7fffffffu ||
		    (uint64_t)v.n_cigar * 4 + v.l_read_name + (((uint64_t)v.l_seq + 1) >> 1) + v.l_seq > rest) return rd_fail("Invalid BAM record.");
		// the record's size comes from the file: take it in steps, so that a damaged size field runs into the end of the data
		// ("BAM file ended prematurely.") instead of into one allocation of gigabytes
		body.clear();
		for (uint32_t done = 0; done < rest;) {
			const uint32_t step = std::min<uint32_t>(rest - done, 16u << 20);
			try { body.resize((size_t)done + step); } catch (const std::bad_alloc &) { return rd_fail("Invalid BAM record."); }
			if (!need(body.data() + done, step)) return false;
			done += step;
		}
		return true;
	}
	std::vector<std::string> names;
private:
	[[noreturn]] void open_fail() { error("Cannot open BAM file '%s'", path_.c_str()); }
	static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
	bool get(uint8_t *dst, size_t n)
	{
		size_t got = 0;
		while (got < n) {
			const long r = bz_->read(dst + got, n - got);
			if (r <= 0) return false;
			got += (size_t)r;
		}
		return true;
	}
	bool rd_fail(const char *msg) { err_ = msg; return false; }
	bool need(uint8_t *dst, size_t n)
	{
		size_t got = 0;
		while (got < n) {
			const long r = bz_->read(dst + got, n - got);
			if (r < 0) return rd_fail("Invalid BAM record.");
			if (r == 0) return rd_fail("BAM file ended prematurely.");
			got += (size_t)r;
		}
		return true;
	}
	bool skip(uint32_t n)                                     // need() without a destination
	{
		size_t got = 0;
		while (got < n) {
			const long r = bz_->skip(n - got);
			if (r < 0) return rd_fail("Invalid BAM record.");
			if (r == 0) return rd_fail("BAM file ended prematurely.");
			got += (size_t)r;
		}
		return true;
	}
	std::string err_;
	std::string path_;
	std::unique_ptr<host::BgzfStream> bz_;
	std::vector<uint8_t> var_;
	std::vector<host::BgzfStream::BamRec> recs_;
};

static const size_t kBatch = 4u << 20;

struct Columns {
	std::vector<uint16_t> flag;
	std::vector<int32_t> tid, mtid, tlen, pos, mpos, end_pos;
	void clear() { flag.clear(); tid.clear(); mtid.clear(); tlen.clear(); pos.clear(); mpos.clear(); end_pos.clear(); }
	void push(const BamCore &c, bool extra)
	{
		flag.push_back(c.flag); tid.push_back(c.tid); mtid.push_back(c.mtid); tlen.push_back(c.tlen);
		if (extra) { pos.push_back(c.pos); mpos.push_back(c.mpos); end_pos.push_back(c.end_pos); }
	}
};

static std::string expand_home(const std::string &path)      // PathArgs::get_path, src/common.rs:28-38
{
	if (!path.empty() && path[0] == '~')
		if (const char *home = getenv("HOME")) return std::string(home) + path.substr(1);
	return path;
}

// ---- sam statistics ----------------------------------------------------------------------------------------------
struct Region { int64_t start, end; };

static int statistics(int argc, char **argv)
{
	std::vector<host::Opt> opts = {{"--on-target", true, false, ""}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_STATS);
	const std::string bam_path = expand_home(pos[0]), targets_path = expand_home(opts[0].value);     // :17-18
	const int64_t max_frag_len = 5000;                                                                 // :19
	host::gpu_warmup();
	auto print_counters = [](const uint64_t counters[3]) {                                             // :109-112
		char buf[256];
		snprintf(buf, sizeof buf, "Total reads: %llu\n", (unsigned long long)counters[0]);
		host::out().write(buf, strlen(buf));
		snprintf(buf, sizeof buf, "Aligned reads: %llu (%s%% of all reads)\n", (unsigned long long)counters[1],
		         host::fmt_pct((double)counters[1] / (double)counters[0] * 100.0).c_str());
		host::out().write(buf, strlen(buf));
		snprintf(buf, sizeof buf, "Duplicate reads: %llu (%s%% of aligned reads)\n", (unsigned long long)counters[2],
		         host::fmt_pct((double)counters[2] / (double)counters[1] * 100.0).c_str());
		host::out().write(buf, strlen(buf));
	};
	// S1 over the FILE (round 6): the compressed bytes cross PCIe, the device inflates the BGZF blocks, walks the records and counts
	// (include/seqkit_hip.h: sk_bam_file_reduce).  It serves well-formed regular files only and says so (handled): everything else —
	// stdin, plain gzip, a file cut short, a record chain that does not verify — is read record by record below, which reports it as
	// the reference does.  (--on-target needs pos / cigar of every record in order: the host's sweep, below.)
	if (targets_path.empty() && bam_path != "-" && !getenv("SEQKIT_HOST_INFLATE")) {
		int handled = 0;
		uint64_t fc[3] = {0, 0, 0};
		const bool trace = getenv("SK_BAMFILE_TRACE") != nullptr;
		auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
		const double t0 = now_ms();
		sk_ctx *c = host::gpu();
		const double t1 = now_ms();
		check(sk_bam_file_reduce(c, bam_path.c_str(), 0, fc, nullptr, nullptr, &handled, nullptr), "sk_bam_file_reduce");
		if (trace) fprintf(stderr, "sam statistics: waited %.1f ms for the device contexts, sk_bam_file_reduce %.1f ms\n", t1 - t0, now_ms() - t1);
		if (handled) { print_counters(fc); return 0; }
	}
	BamStream bam(bam_path);

	std::vector<std::vector<Region>> target_regions;                                                   // :26-54
	if (!targets_path.empty()) {
		fputs("Reading target regions into memory...\n", stderr);
		target_regions.resize(bam.names.size());
		host::LineReader bed(targets_path);
		std::string line;
		for (;;) {
			const bool ok = bed.read_line(line);
			if (bed.bad_utf8()) error("I/O error while reading from file.");
			if (!ok) break;
			const size_t off = host::trim_start_off(line), end = host::trim_end_len(line);
			if (end <= off || line[0] == '#') continue;                                                // :37
			const std::string t = line.substr(off, end - off);
			std::vector<std::string> cols;
			size_t a = 0;
			for (;;) { const size_t b = t.find('\t', a); cols.push_back(t.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
			if (cols.size() < 3) error("Invalid line in BED file %s:\n%s", targets_path.c_str(), line.c_str());
			int tid = -1;
			for (size_t k = 0; k < bam.names.size(); k++) if (bam.names[k] == cols[0]) { tid = (int)k; break; }
			if (tid < 0) error("Chromosome %s is listed in target region BED file, but is not found in BAM file.", cols[0].c_str());
			uint64_t s, e;
			if (!host::parse_uint(cols[1].c_str(), INT64_MAX, s) || !host::parse_uint(cols[2].c_str(), INT64_MAX, e)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (BED)");
			target_regions[tid].push_back({(int64_t)s + 1, (int64_t)e});                               // :44-47
		}
		for (auto &v : target_regions) std::sort(v.begin(), v.end(), [](const Region &a, const Region &b) { return a.start < b.start; });   // :51-53
	}
	const bool on_target = !target_regions.empty();

	uint64_t counters[3] = {0, 0, 0};
	uint64_t total_fragments = 0, on_target_fragments = 0;
	Columns col;
	std::vector<BamCore> chunk;
	bool more = true;
	while (more) {
		col.clear();
		while (col.flag.size() < kBatch && (more = bam.next_chunk(chunk, on_target)))
			for (const BamCore &c : chunk) col.push(c, on_target);
		const int64_t n = (int64_t)col.flag.size();
		if (n == 0) break;
		// S1 on the device: src/sam_statistics.rs:63-69
		check(sk_bam_flag_tlen(host::gpu(), col.flag.data(), nullptr, nullptr, nullptr, n, 0, counters, nullptr, nullptr), "sk_bam_flag_tlen");
		if (!on_target) continue;
		// S2 (--on-target) stays on the host: src/sam_statistics.rs:72-106
		for (int64_t i = 0; i < n; i++) {
			const uint16_t f = col.flag[i];
			if ((f & 0x100) || (f & 0x800)) continue;
			if (f & 0x4) continue;
			int64_t start, end;
			if (f & 0x1) {
				if (f & 0x8) continue;
				if (col.tid[i] != col.mtid[i]) continue;
				if (col.pos[i] > col.mpos[i] || (col.pos[i] == col.mpos[i] && !(f & 0x40))) continue;
				int64_t tl = col.tlen[i];
				if (tl < 0) tl = -tl;
				if (tl > max_frag_len) continue;
				start = (int64_t)col.pos[i] + 1;
				end = start + tl;
			} else {
				start = (int64_t)col.pos[i] + 1;
				end = (int64_t)col.end_pos[i] + 1;
			}
			total_fragments += 1;
			if (col.tid[i] < 0 || (size_t)col.tid[i] >= target_regions.size()) panic("index out of bounds: target_regions[tid]");
			for (const Region &r : target_regions[col.tid[i]]) {
				if (start <= r.end && end >= r.start) { on_target_fragments += 1; break; }
				if (r.start > end) break;
			}
		}
	}
	bam.raise_deferred();
	print_counters(counters);                                                                          // :109-115
	char buf[256];
	if (on_target) {
		snprintf(buf, sizeof buf, "On-target: %s%%\n", host::fmt_pct((double)on_target_fragments / (double)total_fragments * 100.0).c_str());
		host::out().write(buf, strlen(buf));
	}
	return 0;
}

// ---- sam fragment lengths ---------------------------------------------------------------------------------------
static bool frag_keep(uint16_t f, int32_t tid, int32_t mtid, int32_t tlen, uint64_t max_frag, uint64_t &frag)   // :30-38
{
	if ((f & (0x1 | 0x40 | 0x4 | 0x8 | 0x400 | 0x100 | 0x800)) != (0x1 | 0x40)) return false;
	if (tid != mtid) return false;
	const int64_t t = tlen;
	frag = (uint64_t)(t < 0 ? -t : t);
	return frag <= max_frag;
}

static int fragment_lengths(int argc, char **argv)
{
	std::vector<host::Opt> opts = {{"--max-frag-size", true, false, "5000"}, {"--reads", true, false, "Inf"}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 3, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_FRAG);
	uint64_t max_frag = 5000, stop = UINT64_MAX;
	if (!host::parse_uint(opts[0].value.c_str(), UINT64_MAX, max_frag)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (--max-frag-size)");   // :19-20
	if (opts[1].value != "Inf" && !host::parse_uint(opts[1].value.c_str(), UINT64_MAX, stop)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (--reads)");   // :21-25
	// |tlen| is at most 2^31, so bins above that can never be hit; the device interface takes an int32 bin count
	if (max_frag > 200000000ull) error("--max-frag-size above 200000000 is not supported by this build.");
	std::vector<uint64_t> hist(max_frag + 1, 0);                                                       // :27
	uint64_t total = 0;
	host::gpu_warmup();
	// H1 over the file on the device (see statistics()); the --reads=N stop depends on record order and keeps the host's path
	bool by_file = false;
	if (stop == UINT64_MAX && pos[0] != "-" && !getenv("SEQKIT_HOST_INFLATE")) {
		int handled = 0;
		uint64_t ft = 0;
		std::vector<uint64_t> fh(max_frag + 1, 0);
		check(sk_bam_file_reduce(host::gpu(), pos[0].c_str(), (int32_t)max_frag, nullptr, fh.data(), &ft, &handled, nullptr), "sk_bam_file_reduce");
		if (handled) { hist.swap(fh); total = ft; by_file = true; }
	}
	std::unique_ptr<BamStream> bam_p(by_file ? nullptr : new BamStream(pos[0]));                      // :30
	Columns col;
	std::vector<BamCore> chunk;
	bool more = !by_file, stopped = false;
	while (more && !stopped) {
		col.clear();
		while (col.flag.size() < kBatch && (more = bam_p->next_chunk(chunk, false)))
			for (const BamCore &c : chunk) col.push(c, false);
		const int64_t n = (int64_t)col.flag.size();
		if (n == 0) break;
		// H1 on the device: src/sam_fragment_lengths.rs:29-43
		std::vector<uint64_t> bh(max_frag + 1, 0);
		uint64_t bt = 0;
		check(sk_bam_flag_tlen(host::gpu(), col.flag.data(), col.tid.data(), col.mtid.data(), col.tlen.data(), n, (int32_t)max_frag, nullptr, bh.data(), &bt), "sk_bam_flag_tlen");
		if (total + bt < stop) {
			for (size_t i = 0; i <= max_frag; i++) hist[i] += bh[i];
			total += bt;
		} else {
			// the --reads=N stop (:42) falls inside this batch: it depends on input order, so the batch is walked here
			for (int64_t i = 0; i < n; i++) {
				uint64_t frag;
				if (!frag_keep(col.flag[i], col.tid[i], col.mtid[i], col.tlen[i], max_frag, frag)) continue;
				total += 1;
				hist[frag] += 1;
				if (total >= stop) { stopped = true; break; }
			}
		}
	}
	if (!stopped && bam_p) bam_p->raise_deferred();
	char buf[64];
	for (uint64_t size = 1; size < max_frag + 1; size++) {                                             // :45-47
		snprintf(buf, sizeof buf, "%llu\t%llu\n", (unsigned long long)size, (unsigned long long)hist[size]);
		host::out().write(buf, strlen(buf));
	}
	return 0;
}

// ---- sam fragments ---------------------------------------------------------------------------
static const char *USAGE_FRAGMENTS =
	"\nUsage:\n  sam fragments [options] <bam_file>\n\nOptions:\n"
	"  --min-size=N     Minimum fragment size [default: 0]\n"
	"  --max-size=N     Maximum fragment size [default: 5000]\n";

static bool parse_i64(const std::string &s, int64_t &out)          // str::parse::<i64>()
{
	const char *p = s.c_str();
	bool neg = false;
	if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
	uint64_t v;
	if (!host::parse_uint(p, neg ? 9223372036854775808ull : 9223372036854775807ull, v)) return false;
	out = neg ? (int64_t)(0 - v) : (int64_t)v;
	return true;
}

static int fragments(int argc, char **argv)                        // src/sam_fragments.rs:14-43
{
	std::vector<host::Opt> opts = {{"--min-size", true, false, "0"}, {"--max-size", true, false, "5000"}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_FRAGMENTS);
	int64_t min_size, max_size;
	if (!parse_i64(opts[0].value, min_size)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (--min-size)");   // :17
	if (!parse_i64(opts[1].value, max_size)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (--max-size)");   // :18
	host::gpu_warmup();
	BamStream bam(pos[0]);
	Columns col;
	std::vector<BamCore> chunk;
	bool more = true;
	std::vector<uint8_t> bits;
	char buf[128];
	while (more) {
		col.clear();
		while (col.flag.size() < kBatch && (more = bam.next_chunk(chunk, false)))
			for (const BamCore &c : chunk) col.push(c, true);
		const int64_t n = (int64_t)col.flag.size();
		if (n == 0) break;
		bits.assign((size_t)(n + 7) / 8, 0);
		uint64_t kept = 0;
		// the record filter on the device: src/sam_fragments.rs:27-38
		check(sk_bam_fragments(host::gpu(), col.flag.data(), col.tid.data(), col.mtid.data(), col.tlen.data(), n, min_size, max_size, bits.data(), &kept), "sk_bam_fragments");
		if (kept == 0) continue;
		for (int64_t i = 0; i < n; i++) {
			if (!(bits[(size_t)i >> 3] >> (i & 7) & 1)) continue;
			if (col.tid[i] < 0 || (size_t)col.tid[i] >= bam.names.size()) panic("index out of bounds: chr_names[tid]");
			int64_t t = col.tlen[i];
			if (t < 0) t = -t;
			snprintf(buf, sizeof buf, "\t%lld\t%lld\n", (long long)col.pos[i], (long long)col.pos[i] + (long long)t);      // :41
			host::out().write(bam.names[col.tid[i]]);
			host::out().write(buf, strlen(buf));
		}
	}
	bam.raise_deferred();
	return 0;
}

// ---- sam count (second half) --------------------------------------------------------------------
static const char *USAGE_COUNT =
	"\nUsage:\n  sam count [options] <bam_file> <regions.bed>\n\nOptions:\n"
	"  --min-mapq=N      Only count reads with MAPQ \xe2\x89\xa5 threshold [default: 0]\n"
	"  --max-frag-len=N  Maximum allowed DNA fragment length [default: 5000]\n"
	"  --single-end      Count individual reads, rather than DNA fragments\n"
	"  --center          Only count fragments whose center is within a region\n"
	"\n"
	"Counts the number of DNA fragments (or single reads) in the input BAM file\n"
	"that overlap each region described in the input BED file. The BAM file must\n"
	"be position-sorted.\n";

static int count(int argc, char **argv)                            // src/sam_count.rs:20-130
{
	std::vector<host::Opt> opts = {{"--min-mapq", true, false, "0"}, {"--max-frag-len", true, false, "5000"}, {"--single-end", false, false, ""},
	                               {"--center", false, false, ""}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_COUNT);
	uint64_t v;
	if (!host::parse_uint(opts[0].value.c_str(), 255, v)) error("--min-mapq must be an integer between 0 - 255.");       // :23-24
	const uint8_t min_mapq = (uint8_t)v;
	if (!host::parse_uint(opts[1].value.c_str(), 0xffffffffull, v)) error("--max-frag-len must be an integer.");          // :25
	const uint32_t max_frag_len = (uint32_t)v;
	const bool single_end = opts[2].present, count_centers = opts[3].present;                                              // :26-27

	fputs("Reading target regions from BED file...\n", stderr);                                                            // :30
	struct Reg { std::string chr; uint32_t start, end; };
	std::vector<Reg> regions;                                                                                              // read_regions, src/common.rs:198-219
	{
		host::LineReader bed(pos[1]);
		std::string line;
		for (;;) {
			const bool ok = bed.read_line(line);
			if (bed.bad_utf8()) error("I/O error while reading from file.");
			if (!ok) break;
			if (!line.empty() && line[0] == '#') continue;
			const size_t off = host::trim_start_off(line), end = host::trim_end_len(line);
			const std::string t = end > off ? line.substr(off, end - off) : std::string();
			std::vector<std::string> cols;
			size_t a = 0;
			for (;;) { const size_t b = t.find('\t', a); cols.push_back(t.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
			if (cols.size() < 3) error("Invalid region in BED file:\n%s", line.c_str());
			uint64_t s, e;
			if (!host::parse_uint(cols[1].c_str(), 0xffffffffull, s) || !host::parse_uint(cols[2].c_str(), 0xffffffffull, e))
				panic("called `Result::unwrap()` on an `Err` value: ParseIntError (BED)");
			regions.push_back({cols[0], (uint32_t)s, (uint32_t)e});
		}
	}
	fprintf(stderr, "Counting %s...\n", single_end ? "reads" : "DNA fragments");                                            // :34-35
	host::gpu_warmup();
	BamStream bam(pos[0]);                                                                                                 // :36
	for (const std::string &nm : bam.names)                                                                                // :37-38
		if (!host::utf8_valid(reinterpret_cast<const uint8_t *>(nm.data()), nm.size())) panic("called `Result::unwrap()` on an `Err` value: Utf8Error");

	// the regions of every BAM reference (:61-63), handed to the device once
	{
		const int n_chr = (int)bam.names.size();
		std::vector<int32_t> chr_off(n_chr + 1, 0), ridx;
		std::vector<uint32_t> rs, re;
		for (int c = 0; c < n_chr; c++) {
			for (size_t r = 0; r < regions.size(); r++)
				if (regions[r].chr == bam.names[c]) { rs.push_back(regions[r].start); re.push_back(regions[r].end); ridx.push_back((int32_t)r); }
			chr_off[c + 1] = (int32_t)rs.size();
		}
		check(sk_count_set_regions(host::gpu(), n_chr, chr_off.data(), rs.data(), re.data(), ridx.data(), (int64_t)rs.size(),
		                           (int64_t)std::max(rs.size(), regions.size())), "sk_count_set_regions");
	}

	int32_t prev_chr = -1;                                                                                                 // :40-41
	int64_t prev_pos = 0;
	Columns col;
	std::vector<uint8_t> mapq;
	std::vector<BamCore> chunk;
	bool more = true;
	const char *stop = nullptr;                      // an order-dependent error met while reading: raised after the records before it
	int stop_code = 255;
	while (more && !stop) {
		col.clear(); mapq.clear();
		while (col.flag.size() < kBatch && !stop && (more = bam.next_chunk(chunk, single_end))) {
			for (const BamCore &c : chunk) {
				// :46-49 and the order checks :52-73 stay here: they depend on the records before
				if (!((c.flag & 0x4) || (c.flag & 0x400) || (c.flag & 0x100) || (c.flag & 0x800) || c.mapq < min_mapq)) {
					if (c.tid != prev_chr) {
						prev_chr = c.tid;
						if (c.tid < 0 || (size_t)c.tid >= bam.names.size()) { stop = "index out of bounds: chr_names[tid]"; stop_code = 101; break; }
					} else if ((int64_t)c.pos < prev_pos) {
						stop = "Input BAM file is not coordinate sorted."; break;                                              // :70-72
					}
					prev_pos = c.pos;
				}
				col.push(c, true);
				mapq.push_back(c.mapq);
			}
		}
		const int64_t n = (int64_t)col.flag.size();
		if (n == 0) continue;
		check(sk_count_add(host::gpu(), col.flag.data(), mapq.data(), col.tid.data(), col.mtid.data(), col.pos.data(), col.mpos.data(), col.tlen.data(),
		                   single_end ? col.end_pos.data() : nullptr, n, min_mapq, max_frag_len, single_end ? 1 : 0, count_centers ? 1 : 0), "sk_count_add");
	}
	if (stop) { if (stop_code == 101) panic(stop); error("%s", stop); }
	bam.raise_deferred();
	std::vector<uint32_t> region_frags(std::max<size_t>(regions.size(), 1));
	check(sk_count_get(host::gpu(), region_frags.data()), "sk_count_get");
	char buf[32];
	for (size_t r = 0; r < regions.size(); r++) {                                                                          // :128-130
		snprintf(buf, sizeof buf, "%u\n", region_frags[r]);
		host::out().write(buf, strlen(buf));
	}
	return 0;
}

// ---- sam to raw|fasta|fastq -----------------------------------------------------------------
static const char *USAGE_TO =
	"\nUsage:\n"
	"  sam to raw <bam_file> <out_prefix>\n"
	"  sam to fasta <bam_file> <out_prefix>\n"
	"  sam to fastq <bam_file> <out_prefix>\n"
	"  sam to interleaved raw <bam_file>\n"
	"  sam to interleaved fasta <bam_file>\n"
	"  sam to interleaved fastq <bam_file>\n"
	"\n"
	"These commands convert BAM files into FASTQ, FASTA, or raw sequence-per-line\n"
	"format. Both name-sorted and position-sorted BAM files are supported,\n"
	"but memory usage can reach several GB for position-sorted BAM files.\n"
	"\n"
	"Output is written into files whose name is derived based on output prefix\n"
	"and format. For example, with output format FASTQ and prefix \"sample\",\n"
	"paired end reads are written into files sample_1.fq.gz and sample_2.fq.gz,\n"
	"and orphan reads are written into sample.fq.gz.\n";

enum class OutFmt { RAW, FASTA, FASTQ };

// one of the three destinations of write_reads (:92-93): a gzip file, stdout, or io::sink()
struct ReadSink {
	std::unique_ptr<host::GzWriter> gz;
	bool to_stdout = false;
	void write(const char *p, size_t n)
	{
		if (gz) gz->write(p, n);
		else if (to_stdout) host::out().write(p, n);
	}
	void write(const std::string &s) { write(s.data(), s.size()); }
};

static bool char_boundary(const std::string &s, size_t i) { return i == s.size() || (i < s.size() && ((uint8_t)s[i] & 0xC0) != 0x80); }

static void write_read(ReadSink &out, OutFmt format, const std::string &qname, const std::string &seq)       // :138-149
{
	if (format == OutFmt::FASTQ) {
		const size_t seq_len = (seq.size() - 1) / 2;                                                          // :141
		if (!char_boundary(seq, seq_len) || !char_boundary(seq, seq_len + 1)) panic("byte index is not a char boundary");
		out.write("@", 1); out.write(qname); out.write("\n", 1);
		out.write(seq.data(), seq_len); out.write("\n+\n", 3);
		out.write(seq.data() + seq_len + 1, seq.size() - seq_len - 1); out.write("\n", 1);
	} else if (format == OutFmt::FASTA) {
		out.write(">", 1); out.write(qname); out.write("\n", 1); out.write(seq); out.write("\n", 1);
	} else {
		out.write(seq); out.write("\n", 1);
	}
}

// HashMap<Box<str>, Box<str>> (:98-99) whose leftovers are listed in insertion order (the reference's order is arbitrary)
struct PendingReads {
	struct Val { std::string seq; uint64_t order; };
	std::unordered_map<std::string, Val> map;
	uint64_t next = 0;
	void insert(const std::string &qname, std::string &&seq)
	{
		auto it = map.find(qname);
		if (it != map.end()) it->second.seq = std::move(seq);           // HashMap::insert replaces the value
		else map.emplace(qname, Val{std::move(seq), next++});
	}
	std::vector<const std::pair<const std::string, Val> *> in_order() const
	{
		std::vector<const std::pair<const std::string, Val> *> v;
		v.reserve(map.size());
		for (const auto &kv : map) v.push_back(&kv);
		std::sort(v.begin(), v.end(), [](auto a, auto b) { return a->second.order < b->second.order; });
		return v;
	}
};

static ReadSink g_sinks[3];
static void close_sinks() { for (auto &s : g_sinks) if (s.gz) s.gz->close(); }

static int to_reads(int argc, char **argv)
{
	const bool interleaved = argc >= 4 && strcmp(argv[2], "interleaved") == 0;
	const char *fmtw = argv[interleaved ? 3 : 2];
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, interleaved ? 4 : 3, opts, pos, 2) || pos.size() != (interleaved ? 1u : 2u)) error("Invalid arguments.\n%s", USAGE_TO);
	const OutFmt format = !strcmp(fmtw, "raw") ? OutFmt::RAW : !strcmp(fmtw, "fasta") ? OutFmt::FASTA : OutFmt::FASTQ;      // :68-71
	ReadSink &out_1 = g_sinks[0], &out_2 = g_sinks[1], &out_single = g_sinks[2];
	if (interleaved) { out_1.to_stdout = true; out_2.to_stdout = true; }                                    // :74-78
	else {                                                                                                   // :79-86
		const char *ext = format == OutFmt::RAW ? "seq" : format == OutFmt::FASTA ? "fa" : "fq";
		out_1.gz.reset(new host::GzWriter(pos[1] + "_1." + ext + ".gz"));
		out_2.gz.reset(new host::GzWriter(pos[1] + "_2." + ext + ".gz"));
		out_single.gz.reset(new host::GzWriter(pos[1] + "." + ext + ".gz"));
		host::at_exit_flush(close_sinks);
	}
	host::gpu_warmup();
	BamStream bam(pos[0]);                                                                                   // :96
	PendingReads reads_1, reads_2;

	// a batch of primary records: their bases go through sequence() on the device (:31-59), the rest is text
	struct Rec { std::string qname; uint16_t flag; uint32_t l_seq; size_t off4, offq; };
	std::vector<Rec> recs;
	std::vector<uint8_t> body, raw4, rawq, m4, mq, mout;
	std::vector<uint16_t> lens, flags;
	const size_t kBatchBytes = 48u << 20;
	BamCore c;
	BamStream::Var v;
	bool more = true;
	std::string read_seq;
	while (more) {
		recs.clear(); raw4.clear(); rawq.clear();
		uint32_t max_len = 0;
		while ((more = bam.next_full(c, v, body))) {                                                         // :101
			if ((c.flag & 0x100) || (c.flag & 0x800)) continue;                                              // :102
			if (v.l_seq > 65532) error("Read longer than 65532 bases: not supported by this build.");
			Rec r;
			r.qname.assign(reinterpret_cast<const char *>(body.data()), v.l_read_name - 1);                  // rust-htslib qname(): without the final NUL
			r.flag = c.flag;
			r.l_seq = v.l_seq;
			const uint8_t *seq4 = body.data() + v.l_read_name + 4 * (size_t)v.n_cigar, *qual = seq4 + (v.l_seq + 1) / 2;
			r.off4 = raw4.size(); r.offq = rawq.size();
			raw4.insert(raw4.end(), seq4, seq4 + (v.l_seq + 1) / 2);
			rawq.insert(rawq.end(), qual, qual + v.l_seq);
			recs.push_back(std::move(r));
			max_len = std::max(max_len, v.l_seq);
			if ((recs.size() + 1) * (size_t)(((max_len + 7) & ~7u) + 4) * 3 > kBatchBytes) break;
		}
		const size_t n = recs.size();
		if (n == 0) continue;
		// row pitch: a multiple of 8, so that sequence() runs as its eight-bytes-per-thread kernel whatever the read length
		// (reads within 4 bases of the C-ABI's 65 532 keep the multiple of 4)
		const size_t stride8 = std::max<size_t>(8, (max_len + 7) & ~(size_t)7);
		const size_t stride = stride8 <= 65532 ? stride8 : std::max<size_t>(4, (max_len + 3) & ~(size_t)3), stride4 = std::max<size_t>(4, (stride / 2 + 3) & ~(size_t)3);
		m4.assign(n * stride4, 0); mq.assign(n * stride, 0); mout.resize(n * stride);
		lens.resize(n); flags.resize(n);
		for (size_t i = 0; i < n; i++) {
			memcpy(m4.data() + i * stride4, raw4.data() + recs[i].off4, (recs[i].l_seq + 1) / 2);
			memcpy(mq.data() + i * stride, rawq.data() + recs[i].offq, recs[i].l_seq);
			lens[i] = (uint16_t)recs[i].l_seq;
			flags[i] = recs[i].flag;
		}
		check(sk_bam_sequence(host::gpu(), m4.data(), (int)stride4, mq.data(), (int)stride, lens.data(), flags.data(), (int64_t)n, 10 /* :105 */, mout.data()),
		      "sk_bam_sequence");
		for (size_t i = 0; i < n; i++) {
			const Rec &r = recs[i];
			if (!host::utf8_valid(reinterpret_cast<const uint8_t *>(r.qname.data()), r.qname.size())) {     // :104 str::from_utf8(..).unwrap()
				close_sinks();
				panic("called `Result::unwrap()` on an `Err` value: Utf8Error");
			}
			read_seq.assign(reinterpret_cast<const char *>(mout.data() + i * stride), r.l_seq);
			if (format == OutFmt::FASTQ) {                                                                   // :107-112
				read_seq.push_back('|');
				const uint8_t *q = rawq.data() + r.offq;
				// qualities below 95 (all real ones) are one ASCII byte each: add 33 to the whole row at once; a row with a
				// larger value takes the byte-by-byte form of char::from(u8)
				const size_t at = read_seq.size();
				read_seq.resize(at + r.l_seq);
				uint8_t high = 0;
				for (uint32_t k = 0; k < r.l_seq; k++) {
					const uint8_t ch = (uint8_t)(33 + q[k]);                                                 // u8 arithmetic wraps (release build)
					read_seq[at + k] = (char)ch;
					high |= ch;
				}
				if (high & 0x80) {
					read_seq.resize(at);
					for (uint32_t k = 0; k < r.l_seq; k++) {
						const uint8_t ch = (uint8_t)(33 + q[k]);
						if (ch < 0x80) read_seq.push_back((char)ch);
						else { read_seq.push_back((char)(0xC0 | (ch >> 6))); read_seq.push_back((char)(0x80 | (ch & 0x3F))); }   // char::from(u8) as UTF-8
					}
				}
			}
			if (!(r.flag & 0x1)) {                                                                           // :114-115
				write_read(out_single, format, r.qname, read_seq);
			} else if (r.flag & 0x40) {                                                                      // :116-122
				auto it = reads_2.map.find(r.qname);
				if (it != reads_2.map.end()) {
					write_read(out_1, format, r.qname, read_seq);
					write_read(out_2, format, r.qname, it->second.seq);
					reads_2.map.erase(it);
				} else reads_1.insert(r.qname, std::string(read_seq));
			} else if (r.flag & 0x80) {                                                                      // :123-130
				auto it = reads_1.map.find(r.qname);
				if (it != reads_1.map.end()) {
					write_read(out_1, format, r.qname, it->second.seq);
					write_read(out_2, format, r.qname, read_seq);
					reads_1.map.erase(it);
				} else reads_2.insert(r.qname, std::string(read_seq));
			}
		}
	}
	bam.raise_deferred();
	for (const PendingReads *m : {&reads_1, &reads_2})                                                       // :133-137
		for (const auto *kv : m->in_order()) write_read(out_single, format, kv->first, kv->second.seq);
	close_sinks();
	return 0;
}

int main(int argc, char **argv)
{
	// blocks, per-sample strings and gzip jobs are hundreds of KiB each: above glibc's default mmap threshold every one of them was a
	// mapping of its own, faulted in page by page and given back when freed (2.3 s of system time in a demultiplex of 8 M reads).  From
	// the arenas they are recycled.  (SEQKIT_MALLOC_DEFAULT=1: glibc's defaults, for A/B)
	if (!getenv("SEQKIT_MALLOC_DEFAULT")) {
		mallopt(M_MMAP_THRESHOLD, 32 << 20);
		mallopt(M_TRIM_THRESHOLD, 1 << 30);
		mallopt(M_TOP_PAD, 64 << 20);
	}
	int rc = 0;
	auto is = [&](int i, const char *w) { return argc > i && strcmp(argv[i], w) == 0; };
	if (argc >= 2 && is(1, "count")) rc = count(argc, argv);
	else if (argc >= 2 && is(1, "fragments")) rc = fragments(argc, argv);
	else if (argc >= 2 && is(1, "statistics")) rc = statistics(argc, argv);
	else if (argc >= 3 && is(1, "fragment") && is(2, "lengths")) rc = fragment_lengths(argc, argv);
	else if (argc >= 3 && is(1, "to") && (is(2, "raw") || is(2, "fasta") || is(2, "fastq"))) rc = to_reads(argc, argv);
	else if (argc >= 4 && is(1, "to") && is(2, "interleaved") && (is(3, "raw") || is(3, "fasta") || is(3, "fastq"))) rc = to_reads(argc, argv);
	else fprintf(stderr, "%s\n", USAGE_TOP);
	host::out().flush();
	// everything is written and closed: what is left is taking the process apart (static destructors, the HIP runtime's exit handlers,
	// gigabytes of heap) — 0.16 s of a demultiplex of 8 M reads.  That is left to the kernel, as on the error path (host::error);
	// SEQKIT_SLOW_EXIT=1 (and SEQKIT_PROF, whose last lines are printed by destructors) returns from main instead.
	if (!getenv("SEQKIT_SLOW_EXIT") && !getenv("SEQKIT_PROF")) { host::flush_for_exit(); fflush(stdout); fflush(stderr); _exit(rc); }
	return rc;
}
