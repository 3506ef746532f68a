// fasta — the `fasta` binary of the reference for its per-read hot path, with the per-read arithmetic done by
// the MI355X library behind include/seqkit_hip.h.  Same command words, arguments, stdin/stdout/file surface,
// messages and exit codes as the reference (dispatch: src/fasta_main.rs:42-82):
//
//   fasta trim by quality <fastq_file> <min_baseq>                 src/fasta_trim_by_quality.rs:10-50
//   fasta mask by quality <fastq_file> <min_baseq>                 src/fasta_mask_by_quality.rs:11-47
//   fasta add barcode <fastq_file> <barcode_file>                  src/fasta_add_barcode.rs:11-45
//   fasta demultiplex [options] <sample_sheet> <fastq_1> [<fastq_2>]   src/fasta_demultiplex.rs:30-265
//   fasta statistics <fastq_file>                                  src/fasta_statistics.rs:12-51   (device census)
//   fasta gc content <genome.fa> <regions.bed>                     src/fasta_gc_content.rs:17-50
//   the host-only line filters live in fasta_text.cpp
//
// The host owns parsing and I/O and keeps the reference's record-at-a-time ORDER of effects: records are
// gathered into batches (fixed-stride SoA), the batch goes through the C-ABI, and the results are emitted
// in input order; an input error that the reference would hit at record i is raised after records < i have
// been emitted, exactly where the reference would have stopped.
#include <unistd.h>
#include <malloc.h>
#include <algorithm>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_common.h"

using host::error;
using host::panic;

static const char *USAGE_TOP =
	"\nUsage:\n"
	"  fasta check <fasta/fastq>\n"
	"  fasta to raw <fasta/fastq>\n"
	"  fasta add base qualities <fasta> <baseq>\n"
	"  fasta remove base qualities <fastq>\n"
	"  fasta simplify read ids <fastq_file>\n"
	"  fasta interleave <fastq_1> <fastq_2>\n"
	"  fasta deinterleave <interleaved_fastq> <out_prefix>\n"
	"  fasta split into anchors <fastq> <anchor_len>\n"
	"  fasta trim <fastq_file>\n"
	"  fasta trim by quality <fastq_file> <min_baseq>\n"
	"  fasta mask by quality <fastq_file> <min_baseq>\n"
	"  fasta gc content <genome.fa> <regions.bed>\n"
	"  fasta add barcode <fastq_file> <barcode_file> <barcode_format>\n"
	"  fasta extract dual umi <interleaved_fastq>\n"
	"  fasta convert basespace <fastq_file>\n"
	"  fasta demultiplex <sample_sheet> <fastq_1> <fastq_2>\n"
	"  fasta demultiplex spe <sample_sheet> <fastq_1> <fastq_2>\n"
	"  fasta statistics <fastq_file>\n";
static const char *USAGE_TRIM = "\nUsage:\n  fasta trim by quality <fastq_file> <min_baseq>\n";
static const char *USAGE_MASK = "\nUsage:\n  fasta mask by quality <fastq_file> <min_baseq>\n";
static const char *USAGE_ADDBC = "\nUsage:\n  fasta add barcode <fastq_file> <barcode_file>\n";
static const char *USAGE_STATS = "\nUsage:\n  fasta statistics <fastq_file>\n";
static const char *USAGE_DEMUX =
	"\nUsage:\n"
	"  fasta demultiplex [options] <sample_sheet> <fastq_1> [<fastq_2>]\n"
	"\n"
	"Options:\n"
	"  --parallel      Use pigz (parallel gzip) for compression\n"
	"  --index1=FASTQ  Path to FASTQ file containing the first index (optional)\n"
	"  --index2=FASTQ  Path to FASTQ file containing the second index (optional)\n"
	"  --dry-run=N     Analyze N reads and generate table of indexes found in the run\n"
	"\n"
	"Splits a pooled FASTQ file into multiple individual FASTQ files, based on a\n"
	"sample sheet. Each read in the pooled FASTQ file must carry a BC:xxxxxxxx\n"
	"field in its header.\n";

static const size_t kBatchBytes = 192u << 20;      // SoA bytes per batch
static const size_t kBatchRecords = 1u << 18;
static const size_t kMaxRow = 65535;               // len is u16 at the C-ABI

static void check(sk_ctx *ctx, int rc, const char *what)
{
	if (rc != SK_OK) error("%s failed: %s", what, sk_last_error(ctx));
}
static void check(int rc, const char *what) { check(host::gpu(), rc, what); }

static uint8_t parse_min_baseq(const std::string &s)
{
	uint64_t v;
	if (!host::parse_uint(s.c_str(), 255, v)) panic("called `Result::unwrap()` on an `Err` value: ParseIntError (min_baseq)");
	return (uint8_t)v;
}

// a batch of rows packed at a fixed stride, in pinned memory (what the device copies from / into by DMA)
struct Matrix {
	uint8_t *data = nullptr;
	int stride = 1;
	void alloc(size_t nrows, size_t row_bytes, host::PinnedArena &arena)
	{
		stride = (int)std::max<size_t>(row_bytes, 1);
		data = arena.take(nrows * (size_t)stride + 16);
	}
	void put(size_t r, const char *p, size_t n)              // row r = p[0..n), zero padded
	{
		uint8_t *d = data + r * (size_t)stride;
		memcpy(d, p, n);
		if (n < (size_t)stride) memset(d + n, 0, (size_t)stride - n);
	}
	void pack(const std::vector<host::Line> &rows, host::PinnedArena &arena)
	{
		size_t mx = 1;
		for (const host::Line &l : rows) mx = std::max(mx, l.n);
		alloc(rows.size(), mx, arena);
		for (size_t r = 0; r < rows.size(); r++) put(r, rows[r].p, rows[r].n);
	}
};

static bool line_utf8_ok(const host::Line &l) { return host::utf8_valid(reinterpret_cast<const uint8_t *>(l.p), l.n); }

// ---------------------------------------------------------------------------------------------------------
// fasta trim by quality — one block of whole records (src/fasta_trim_by_quality.rs:19-49)
// ---------------------------------------------------------------------------------------------------------
struct TrimRec { host::Line header, seq, qual; size_t qual_line; bool body; };

static void trim_block(const char *data, size_t n, uint8_t min_baseq, host::BlockResult &res)
{
	host::BlockLines bl(data, n);
	std::vector<TrimRec> recs;
	std::vector<host::Line> rows;
	recs.reserve(n / 300 + 4);
	for (;;) {
		TrimRec r;
		r.body = false;
		r.header = bl.next();                                               // :19
		if (r.header.n == 0) break;
		if (!line_utf8_ok(r.header)) { res.err = "I/O error while reading from file."; break; }
		if (r.header.p[0] != '@') { res.err = "Invalid FASTQ format encountered."; break; }                  // :20-22
		// :23 prints the header before the rest of the record is read
		r.seq = bl.next();                                                  // :24  (past the end: an empty string)
		host::Line plus = bl.next();                                        // :25
		r.qual = bl.next();                                                 // :26
		if (!line_utf8_ok(r.seq) || !line_utf8_ok(plus) || !line_utf8_ok(r.qual)) {
			recs.push_back(r);                                              // its header is out already
			res.err = "I/O error while reading from file.";
			break;
		}
		r.body = true;
		r.qual_line = r.qual.n;
		r.qual.n = host::trim_end_len(r.qual.p, r.qual.n);                  // :31  n = qual.trim_end().len()
		if (r.qual.n > kMaxRow) { res.err = "Read longer than 65535 bases: not supported by this build."; break; }
		rows.push_back(r.qual);
		recs.push_back(r);
	}
	// T1 through the C-ABI: src/fasta_trim_by_quality.rs:28-42
	host::PinnedArena arena(n / 2 + 65536);
	uint16_t *lowest_k = arena.take_n<uint16_t>(rows.size() + 1);
	if (!rows.empty()) {
		Matrix q;
		q.pack(rows, arena);
		uint16_t *len16 = arena.take_n<uint16_t>(rows.size());
		for (size_t i = 0; i < rows.size(); i++) len16[i] = (uint16_t)rows[i].n;
		host::GpuLease g;
		check(g.ctx(), sk_trim_by_quality(g.ctx(), q.data, len16, q.stride, (int64_t)rows.size(), min_baseq, lowest_k), "sk_trim_by_quality");
	}
	res.out.reserve(n + 16);
	size_t k = 0;
	for (const TrimRec &r : recs) {
		res.out.append(r.header.p, r.header.n);
		if (!r.body) break;
		const size_t lk = lowest_k[k++];
		if (lk == 0) {                                                      // :44-45
			res.out.append("N\n+\n!\n", 6);
		} else {                                                            // :47 — byte slices of seq and qual
			const char *why = nullptr;
			if (lk > r.seq.n) why = "byte index out of range of `seq`";
			else if (lk < r.seq.n && ((uint8_t)r.seq.p[lk] & 0xC0) == 0x80) why = "byte index is not a char boundary (seq)";
			else if (lk < r.qual_line && ((uint8_t)r.qual.p[lk] & 0xC0) == 0x80) why = "byte index is not a char boundary (qual)";
			if (why) { res.err = why; res.err_code = 101; break; }
			res.out.append(r.seq.p, lk);
			res.out.append("\n+\n", 3);
			res.out.append(r.qual.p, lk);
			res.out.push_back('\n');
		}
	}
}

static int trim_by_quality(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_TRIM);
	{ host::LineReader probe(pos[0]); }                                     // :12  FileReader::new comes before the parse (error order)
	const uint8_t min_baseq = parse_min_baseq(pos[1]);                      // :13
	host::gpu_warmup();
	host::run_block_pipeline(pos[0], 4, [min_baseq](const char *d, size_t n, bool, host::BlockResult &res) { trim_block(d, n, min_baseq, res); });
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// fasta mask by quality — one block of whole records (src/fasta_mask_by_quality.rs:20-46)
// ---------------------------------------------------------------------------------------------------------
struct MaskRec { host::Line header, seq, qual; bool ascii; size_t row; };

static size_t u8len(uint8_t b) { return b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : 4; }
static uint32_t u8cp(const uint8_t *p, size_t l)
{
	if (l == 1) return p[0];
	if (l == 2) return ((uint32_t)(p[0] & 0x1F) << 6) | (p[1] & 0x3F);
	if (l == 3) return ((uint32_t)(p[0] & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
	return ((uint32_t)(p[0] & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
}

static void mask_block(const char *data, size_t n, uint8_t min_baseq, host::BlockResult &res)
{
	host::BlockLines bl(data, n);
	std::vector<MaskRec> recs;
	std::vector<host::Line> srows, qrows;
	std::vector<std::string> cseq, cqual;                                   // per-char rows of records with multi-byte chars
	cseq.reserve(16); cqual.reserve(16);
	recs.reserve(n / 300 + 4);
	for (;;) {
		MaskRec r;
		r.header = bl.next();                                               // :20
		if (r.header.n == 0) break;
		if (!line_utf8_ok(r.header)) { res.err = "I/O error while reading from file."; break; }
		if (r.header.p[0] != '@') { res.err = "Invalid FASTQ format encountered."; break; }                  // :21-23
		r.seq = bl.next();                                                  // :28
		host::Line plus = bl.next();                                        // :29
		r.qual = bl.next();                                                 // :30
		if (!line_utf8_ok(r.seq) || !line_utf8_ok(plus) || !line_utf8_ok(r.qual)) { res.err = "I/O error while reading from file."; break; }
		if (r.seq.n && r.seq.p[r.seq.n - 1] == '\n') r.seq.n--;             // :32
		if (r.qual.n && r.qual.p[r.qual.n - 1] == '\n') r.qual.n--;         // :33
		if (r.seq.n != r.qual.n) { res.err = "Read sequence and base qualities are of different length."; break; }   // :35-37
		if (r.seq.n > kMaxRow) { res.err = "Read longer than 65535 bases: not supported by this build."; break; }
		r.ascii = host::is_ascii(r.seq.p, r.seq.n) && host::is_ascii(r.qual.p, r.qual.n);
		recs.push_back(r);
	}
	// M1 through the C-ABI: src/fasta_mask_by_quality.rs:40-43.  The reference zips chars() and takes `qual as u8` (the
	// low byte of the code point).  ASCII records go as they are; for a record with multi-byte characters the host only
	// does the text part — one row element per CHAR (a placeholder base, the low byte of the quality char) — so the
	// threshold arithmetic still runs on the device, and the flagged positions are mapped back.
	size_t nspecial = 0;
	for (const MaskRec &r : recs) nspecial += r.ascii ? 0 : 1;
	cseq.reserve(nspecial); cqual.reserve(nspecial);
	for (MaskRec &r : recs) {
		r.row = srows.size();
		if (r.ascii) { srows.push_back(r.seq); qrows.push_back(r.qual); continue; }
		cseq.emplace_back(); cqual.emplace_back();
		const uint8_t *sp = reinterpret_cast<const uint8_t *>(r.seq.p), *qp = reinterpret_cast<const uint8_t *>(r.qual.p);
		size_t a = 0, b = 0;
		while (a < r.seq.n && b < r.qual.n) {
			const size_t la = u8len(sp[a]), lb = u8len(qp[b]);
			cseq.back().push_back('.');
			cqual.back().push_back((char)(uint8_t)u8cp(qp + b, lb));
			a += la; b += lb;
		}
		srows.push_back({cseq.back().data(), cseq.back().size()});
		qrows.push_back({cqual.back().data(), cqual.back().size()});
	}
	host::PinnedArena arena(n + 65536);
	Matrix s, q;
	if (!srows.empty()) {
		s.pack(srows, arena);
		q.pack(qrows, arena);
		uint16_t *len16 = arena.take_n<uint16_t>(srows.size());
		for (size_t i = 0; i < srows.size(); i++) len16[i] = (uint16_t)srows[i].n;
		host::GpuLease g;
		check(g.ctx(), sk_mask_by_quality(g.ctx(), s.data, q.data, len16, s.stride, (int64_t)srows.size(), min_baseq), "sk_mask_by_quality");
	}
	res.out.reserve(n + 16);
	for (const MaskRec &r : recs) {
		res.out.append(r.header.p, r.header.n);                             // :25-26
		const char *row = reinterpret_cast<const char *>(s.data) + r.row * (size_t)s.stride;
		if (r.ascii) {
			res.out.append(row, r.seq.n);
		} else {
			const uint8_t *sp = reinterpret_cast<const uint8_t *>(r.seq.p);
			size_t a = 0;
			for (size_t c = 0; c < srows[r.row].n; c++) {
				const size_t la = u8len(sp[a]);
				if (row[c] == 'N') res.out.push_back('N');
				else res.out.append(r.seq.p + a, la);
				a += la;
			}
		}
		res.out.append("\n+\n", 3);                                          // :44
		res.out.append(r.qual.p, r.qual.n);
		res.out.push_back('\n');
	}
}

static int mask_by_quality(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_MASK);
	{ host::LineReader probe(pos[0]); }                                     // :13
	const uint8_t min_baseq = parse_min_baseq(pos[1]);                      // :14
	host::gpu_warmup();
	host::run_block_pipeline(pos[0], 4, [min_baseq](const char *d, size_t n, bool, host::BlockResult &res) { mask_block(d, n, min_baseq, res); });
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// fasta add barcode — no arithmetic: pure stream work, restated statement by statement
// ---------------------------------------------------------------------------------------------------------
static int add_barcode(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 3, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_ADDBC);
	host::LineReader fq(pos[0]), bf(pos[1]);
	// lines are looked at where the readers hold them (next_line): only the barcode, which outlives its record when the
	// barcode file runs out, is copied
	std::string barcode;
	auto nx = [](host::LineReader &r, const char *&p, size_t &n) {
		const bool ok = r.next_line(p, n);
		if (r.bad_utf8()) error("I/O error while reading from file.");
		return ok;
	};
	const char *p;
	size_t n;
	for (;;) {
		const bool hb = nx(bf, p, n);                                       // :20
		const char b0 = hb ? p[0] : '\0';
		if (b0 == '@') { if (nx(bf, p, n)) barcode.assign(p, n); else barcode.clear(); nx(bf, p, n); nx(bf, p, n); }   // :21-24
		else if (b0 == '>') { if (nx(bf, p, n)) barcode.assign(p, n); else barcode.clear(); }   // :25-27  (exhausted file: `barcode` keeps its value)
		if (!nx(fq, p, n)) break;                                           // :29-31
		const char h0 = p[0];
		host::out().write(p, host::trim_end_len(p, n));                     // :33
		host::out().write(" BC:", 4);
		host::out().write(barcode.data(), host::trim_end_len(barcode));
		host::out().write("\n", 1);
		if (h0 == '@') { for (int k = 0; k < 3; k++) if (nx(fq, p, n)) host::out().write(p, n); }   // :35-38
		else if (h0 == '>') { if (nx(fq, p, n)) host::out().write(p, n); }                         // :39-40
		else error("Invalid FASTQ line:\n%s", std::string(p, n).c_str());        // :41-43 (the view still holds the header: nothing was read after it)
	}
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// fasta demultiplex
// ---------------------------------------------------------------------------------------------------------
struct Sample {
	std::string name, barcode;
	std::unique_ptr<host::GzWriter> out[2];
	uint64_t total_reads = 0;
};

// ---------------------------------------------------------------------------------------------------------
// f3: barcode census — the HashMap<String, u64> of src/fasta_statistics.rs:19,26 and src/fasta_demultiplex.rs:110,193
// lives on the device (sk_census_*); barcodes the device table cannot key (longer than 31 characters, or bytes outside
// ACGTNacgtn+, which only index files can bring) are counted here.
// ---------------------------------------------------------------------------------------------------------
static const int kCensusStride = 32;
static const size_t kCensusMaxLen = 31;

static bool census_keyable(const char *p, size_t n)
{
	if (n > kCensusMaxLen) return false;
	for (size_t i = 0; i < n; i++) {
		switch (p[i]) {
		case 'A': case 'C': case 'G': case 'T': case 'N': case 'a': case 'c': case 'g': case 't': case 'n': case '+': break;
		default: return false;
		}
	}
	return true;
}

struct CensusEnt { std::string label; uint64_t count; int64_t first; };

struct HostCensus {                                  // what the device table cannot key
	std::unordered_map<std::string, size_t> idx;
	std::vector<CensusEnt> ents;
	void add(const std::string &bc, int64_t first, uint64_t count = 1)
	{
		auto it = idx.find(bc);
		if (it == idx.end()) { idx.emplace(bc, ents.size()); ents.push_back({bc, count, first}); }
		else { ents[it->second].count += count; ents[it->second].first = std::min(ents[it->second].first, first); }
	}
};

// Every barcode that can be among the `need` most frequent ones, in first-seen order.  A small census is fetched
// whole; from a large one only the barcodes whose count reaches the power of two that still leaves `need` of them.
static std::vector<CensusEnt> census_fetch(const HostCensus &hc, size_t need, int64_t dev_first_mul = 1)
{
	std::vector<CensusEnt> all;
	sk_ctx *ctx = host::gpu();
	uint64_t st[4] = {0, 0, 0, 0};
	check(sk_census_stats(ctx, st), "sk_census_stats");
	uint64_t min_count = 1, expect = st[0];
	if (st[0] > (1u << 20)) {
		uint64_t hist[64];
		check(sk_census_count_hist(ctx, hist), "sk_census_count_hist");
		uint64_t above = 0;
		for (int b = 63; b >= 0; b--) {
			above += hist[b];
			if (above >= need || b == 0) { min_count = 1ull << b; expect = above; break; }
		}
	}
	std::vector<sk_census_entry> dev(expect);
	uint64_t total = 0;
	check(sk_census_entries(ctx, min_count, dev.data(), dev.size(), &total), "sk_census_entries");
	if (total != expect) error("census changed while it was read (%llu != %llu).", (unsigned long long)total, (unsigned long long)expect);
	all.reserve(dev.size() + hc.ents.size());
	for (const sk_census_entry &e : dev) all.push_back({e.barcode, e.count, e.first_row * dev_first_mul});
	for (const CensusEnt &e : hc.ents) all.push_back(e);
	std::stable_sort(all.begin(), all.end(), [](const CensusEnt &a, const CensusEnt &b) { return a.first < b.first; });
	return all;
}

// `entries.sort_by_key(|x| x.1); entries.reverse(); for ... in &entries[0..100]` (src/fasta_statistics.rs:45-50,
// src/fasta_demultiplex.rs:255-260).  The reference's entries come out of a HashMap in arbitrary order, so the order
// among equal counts is unspecified there; here they start in first-seen order.  The reference panics when there are
// fewer than 100 entries; this build prints the entries there are (documented deviation).
static void print_most_frequent(std::vector<CensusEnt> &ents)
{
	std::stable_sort(ents.begin(), ents.end(), [](const CensusEnt &a, const CensusEnt &b) { return a.count < b.count; });
	std::reverse(ents.begin(), ents.end());
	const size_t lim = std::min<size_t>(100, ents.size());
	for (size_t i = 0; i < lim; i++) {
		char buf[64];
		snprintf(buf, sizeof buf, ": %llu\n", (unsigned long long)ents[i].count);
		host::out().write("- ", 2); host::out().write(ents[i].label); host::out().write(buf, strlen(buf));
	}
}

// ---------------------------------------------------------------------------------------------------------
// fasta statistics (src/fasta_statistics.rs:12-51)
// ---------------------------------------------------------------------------------------------------------
static int statistics(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_STATS);
	host::LineReader fastq(pos[0]);                                         // :14
	host::gpu_warmup();
	check(sk_census_reset(host::gpu()), "sk_census_reset");
	size_t kBatch = 1u << 20;
	if (const char *e = getenv("SEQKIT_BLOCK_RECORDS")) kBatch = std::max<size_t>(1, (size_t)atoll(e));     // tests use tiny batches
	std::vector<uint8_t> rows(kBatch * kCensusStride, 0);
	size_t nrows = 0;
	int64_t row_base = 0;                                                   // census rows handed to the device so far
	HostCensus longs;
	auto flush = [&]() {
		if (nrows == 0) return;
		check(sk_census_add(host::gpu(), rows.data(), kCensusStride, (int)kCensusMaxLen, (int64_t)nrows, nullptr, row_base), "sk_census_add");
		memset(rows.data(), 0, nrows * kCensusStride);
		row_base += (int64_t)nrows;
		nrows = 0;
	};
	uint64_t total_records = 0;
	std::string line;
	auto read = [&](std::string &l) {                                      // FileReader::read_line (src/common.rs:106-112)
		const bool ok = fastq.read_line(l);
		if (fastq.bad_utf8()) error("I/O error while reading from file.");
		return ok;
	};
	auto skip_line = [&]() {                                                // the same read, the line left where the reader holds it
		const char *p;
		size_t n;
		fastq.next_line(p, n);
		if (fastq.bad_utf8()) error("I/O error while reading from file.");
	};
	while (read(line)) {                                                    // :22
		size_t st = 0, en = 0;
		if (host::find_bc_field(line, st, en, false)) {                     // :24-27
			const char *bc = line.data() + st + 4;
			const size_t n = en - st - 4;
			if (n <= kCensusMaxLen) {
				memcpy(rows.data() + nrows * kCensusStride, bc, n);
				if (++nrows == kBatch) flush();
			} else {
				// between the device rows before and after it: device row i is first-seen at 2i (census_fetch below)
				longs.add(std::string(bc, n), 2 * (row_base + (int64_t)nrows) - 1);
			}
		}
		if (line[0] == '@') { for (int k = 0; k < 3; k++) skip_line(); }    // :30-31
		else if (line[0] == '>') skip_line();                               // :32-33
		else error("Invalid FASTQ header:\n%s", line.c_str());              // :34-36
		total_records += 1;                                                 // :38
	}
	flush();
	char buf[96];
	snprintf(buf, sizeof buf, "Total sequence records: %llu\n", (unsigned long long)total_records);       // :41
	host::out().write(buf, strlen(buf));
	host::out().write("Most frequent sample barcodes:\n");                  // :43
	std::vector<CensusEnt> ents = census_fetch(longs, 100, 2);
	print_most_frequent(ents);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// fasta gc content <genome.fa> <regions.bed> (src/fasta_gc_content.rs:17-50).  The genome goes to the device once; the
// two byte counts of every region (:44-45) come back per batch of BED lines.  The FASTA reader is rust-bio 0.19's
// (`bio::io::fasta::Reader::records`, not in the reference tree), restated from its published behaviour: a record is a
// line starting with '>' and the following lines up to the next '>' line, each with its trailing whitespace removed; the
// id is the header up to its first space.
// ---------------------------------------------------------------------------------------------------------
static const char *USAGE_GC =
	"\nUsage:\n  fasta gc content <genome.fa> <regions.bed>\n\nDescription:\n"
	"Calculates the GC content percentage of FASTA file regions listed in the input\n"
	"BED file. Ambiguous N nucleotides are omitted from both the numerator and the\n"
	"denominator.\n";

static int gc_content(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 3, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_GC);
	fputs("Reading reference genome into memory...\n", stderr);                         // :22
	FILE *fa = fopen(pos[0].c_str(), "rb");                                             // :23 File::open: no gzip, no "-"
	if (!fa) error("Input FASTA file %s could not be read.", pos[0].c_str());
	host::gpu_warmup();
	std::string genome;                                                                 // all sequences, back to back
	std::unordered_map<std::string, std::pair<int64_t, int64_t>> chrs;                  // id -> (offset, length)   :25-29
	{
		std::string line;
		char *buf = nullptr;
		size_t cap = 0;
		auto read_line = [&](std::string &l) {                                          // BufRead::read_line: appends, UTF-8 or Err
			const ssize_t r = getline(&buf, &cap, fa);
			if (r <= 0) return;
			if (!host::utf8_valid(reinterpret_cast<const uint8_t *>(buf), (size_t)r)) panic("called `Result::unwrap()` on an `Err` value: stream did not contain valid UTF-8");
			l.append(buf, (size_t)r);
		};
		for (;;) {
			if (line.empty()) { read_line(line); if (line.empty()) break; }
			if (line[0] != '>') panic("called `Result::unwrap()` on an `Err` value: Expected > at record start.");       // :27 entry.unwrap()
			const size_t hend = host::trim_end_len(line);
			const size_t sp = line.find(' ', 1);
			const std::string id = line.substr(1, (sp == std::string::npos || sp > hend ? hend : sp) - 1);
			const int64_t off = (int64_t)genome.size();
			for (;;) {
				line.clear();
				read_line(line);
				if (line.empty() || line[0] == '>') break;
				genome.append(line.data(), host::trim_end_len(line));
			}
			chrs[id] = {off, (int64_t)genome.size() - off};                             // HashMap::insert: a repeated id replaces
		}
		free(buf);
		fclose(fa);
	}
	check(sk_gc_set_genome(host::gpu(), reinterpret_cast<const uint8_t *>(genome.data()), (int64_t)genome.size()), "sk_gc_set_genome");

	host::LineReader bed(pos[1]);                                                       // :31
	std::vector<int64_t> starts, lens;
	std::vector<uint64_t> gc, total;
	std::string line, stop_msg;
	int stop_code = 0;
	auto flush = [&]() {
		if (starts.empty()) return;
		gc.assign(starts.size(), 0); total.assign(starts.size(), 0);
		check(sk_gc_count(host::gpu(), starts.data(), lens.data(), (int64_t)starts.size(), gc.data(), total.data()), "sk_gc_count");
		char out[96];
		for (size_t i = 0; i < starts.size(); i++) {
			const float ratio = (float)gc[i] / (float)total[i];                         // :46 `as f32`, `{:.3}`
			if (total[i] == 0) snprintf(out, sizeof out, "%llu\t%llu\tNaN\n", (unsigned long long)gc[i], (unsigned long long)total[i]);
			else snprintf(out, sizeof out, "%llu\t%llu\t%.3f\n", (unsigned long long)gc[i], (unsigned long long)total[i], (double)ratio);
			host::out().write(out, strlen(out));
		}
		starts.clear(); lens.clear();
	};
	for (;;) {
		const bool ok = bed.read_line(line);
		if (bed.bad_utf8()) { stop_msg = "I/O error while reading from file."; stop_code = 255; break; }
		if (!ok) break;
		const size_t off = host::trim_start_off(line), end = host::trim_end_len(line);                       // :34 line.trim().split('\t')
		const std::string t = end > off ? line.substr(off, end - off) : std::string();
		std::vector<std::string> cols;
		size_t a = 0;
		for (;;) { const size_t b = t.find('\t', a); cols.push_back(t.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
		if (cols.size() < 3) fprintf(stderr, "WARNING: Input BED file contains line with less than 3 columns:\n%s\n\n", line.c_str());   // :35-37
		auto it = chrs.find(cols[0]);                                                   // :39
		if (it == chrs.end()) continue;
		if (cols.size() < 2) { stop_msg = "index out of bounds: the len is 1 but the index is 1"; stop_code = 101; break; }
		uint64_t start, stop;
		if (!host::parse_uint(cols[1].c_str(), UINT64_MAX, start)) { stop_msg = "Invalid region:\n" + line + "\n"; stop_code = 255; break; }       // :40
		if (cols.size() < 3) { stop_msg = "index out of bounds: the len is 2 but the index is 2"; stop_code = 101; break; }
		if (!host::parse_uint(cols[2].c_str(), UINT64_MAX, stop)) { stop_msg = "Invalid region:\n" + line + "\n"; stop_code = 255; break; }        // :41
		if (start > stop || stop > (uint64_t)it->second.second) { stop_msg = "Invalid region:\n" + line + "\n"; stop_code = 255; break; }         // :42 get(start..stop)
		starts.push_back(it->second.first + (int64_t)start);
		lens.push_back((int64_t)(stop - start));
		if (starts.size() >= (1u << 16)) flush();
	}
	flush();
	if (stop_code == 101) panic(stop_msg.c_str());
	if (stop_code) error("%s", stop_msg.c_str());
	return 0;
}

// ---- one block of clusters (all input files cut at the same record count) -> per-sample text -----------------------
struct DemuxCfg {
	std::vector<Sample> *samples;
	size_t barcode_len;
	bool paired_end;
	int nindex;
	bool do_mask, do_trim;
	uint8_t q;
	bool dry_run;
};

struct DemuxOut {
	size_t nclusters = 0;                          // clusters the reference would have counted (total_reads) in this block
	uint64_t identified = 0;
	std::vector<uint64_t> per_sample;
	std::string warn;                              // stderr text, in cluster order
	std::vector<std::string> out1, out2;           // per sample
	HostCensus extras;                             // dry run: unmatched barcodes the device census cannot key
	std::string err;
	int err_code = 255;
};

struct ClusterV {
	host::Line h, l2, l3, l4, m2[4];
	size_t bc_st = 0, bc_en = 0;                   // header mode: the " BC:..." field inside h
	std::string barcode;
};

static std::vector<Sample> *g_samples = nullptr;
static void close_outputs()
{
	if (!g_samples) return;
	for (auto &s : *g_samples) for (auto &o : s.out) if (o) o->flush_tail();        // all the tails first: one batch for the pool
	for (auto &s : *g_samples) for (auto &o : s.out) if (o) o->close();
}

static inline size_t strip_nl(const host::Line &l) { return l.n - ((l.n && l.p[l.n - 1] == '\n') ? 1 : 0); }

static void demux_block(const DemuxCfg &cfg, const host::Bytes *blk /* fastq1, fastq2, index1, index2 */, uint64_t base /* clusters before this block */, DemuxOut &res)
{
	std::vector<Sample> &samples = *cfg.samples;
	const int S = (int)samples.size();
	const size_t L = cfg.barcode_len;
	const bool fused = cfg.do_mask || cfg.do_trim;
	res.per_sample.assign(S, 0);
	if (!cfg.dry_run) { res.out1.resize(S); if (cfg.paired_end) res.out2.resize(S); }
	host::BlockLines f1(blk[0].data(), blk[0].size()), f2(blk[1].data(), blk[1].size());
	host::BlockLines ix[2] = {host::BlockLines(blk[2].data(), blk[2].size()), host::BlockLines(blk[3].data(), blk[3].size())};
	auto bad = [&](const host::Line &l) { return !line_utf8_ok(l); };

	// ---- gather (src/fasta_demultiplex.rs:117-150; all lines of a cluster are consumed here, :239-246 does it for
	// unassigned reads too)
	std::vector<ClusterV> cl;
	cl.reserve(blk[0].size() / 200 + 4);
	for (;;) {
		ClusterV c;
		c.h = f1.next();                                                    // :117
		if (c.h.n == 0) break;
		if (bad(c.h)) { res.err = "I/O error while reading from file."; break; }
		if (c.h.p[0] != '@') { res.err = "Invalid FASTQ header line:\n" + std::string(c.h.p, c.h.n); break; }   // :118-120
		bool stop = false;
		if (cfg.nindex > 0) {                                               // :126-136
			for (int f = 0; f < cfg.nindex && !stop; f++) {
				if (!c.barcode.empty()) c.barcode += "+";
				host::Line a = ix[f].next();
				if (bad(a)) { res.err = "I/O error while reading from file."; stop = true; break; }
				if (a.n == 0 || a.p[0] != '@') { res.err = "assertion failed: line.starts_with('@')"; res.err_code = 101; stop = true; break; }
				host::Line b = ix[f].next();
				if (bad(b)) { res.err = "I/O error while reading from file."; stop = true; break; }
				c.barcode.append(b.p, host::trim_end_len(b.p, b.n));
				host::Line d = ix[f].next();
				if (bad(d)) { res.err = "I/O error while reading from file."; stop = true; break; }
				if (d.n == 0 || d.p[0] != '+') { res.err = "assertion failed: line.starts_with('+')"; res.err_code = 101; stop = true; break; }
				host::Line e = ix[f].next();
				if (bad(e)) { res.err = "I/O error while reading from file."; stop = true; break; }
			}
		} else {                                                            // :137-146
			const std::string hs(c.h.p, c.h.n);
			if (!host::find_bc_field(hs, c.bc_st, c.bc_en)) { res.err = "No BC:xxxx field found."; stop = true; }
			else c.barcode.assign(hs, c.bc_st + 4, c.bc_en - c.bc_st - 4);
		}
		if (stop) break;
		if (c.barcode.size() != L) {                                        // :148-150
			char buf[512];
			snprintf(buf, sizeof buf, "Sequenced barcode %s is of different length (%zu nt) than barcodes in the sample sheet (%zu nt).",
			         c.barcode.c_str(), c.barcode.size(), L);
			res.err = buf;
			break;
		}
		c.l2 = f1.next(); c.l3 = f1.next(); c.l4 = f1.next();
		if (bad(c.l2) || bad(c.l3) || bad(c.l4)) { res.err = "I/O error while reading from file."; break; }
		if (cfg.paired_end) {
			for (int k = 0; k < 4; k++) c.m2[k] = f2.next();
			if (bad(c.m2[0]) || bad(c.m2[1]) || bad(c.m2[2]) || bad(c.m2[3])) { res.err = "I/O error while reading from file."; break; }
		}
		cl.push_back(std::move(c));
	}

	// ---- D1+D2+D3 (and, fused, M1 + T1 of both mates) for the block through the C-ABI ------------------------------
	const size_t nb = cl.size();
	res.nclusters = nb;
	size_t hint = nb * (L + 16) + 65536;
	if (fused) hint += 3 * (blk[0].size() + blk[1].size()) / 2;
	host::PinnedArena arena(hint);
	int32_t *assign = arena.take_n<int32_t>(nb + 1);
	for (size_t i = 0; i < nb; i++) assign[i] = SK_ASSIGN_NONE;             // empty sheet: lowest_diff stays usize::MAX
	uint8_t *lowest = arena.take_n<uint8_t>(nb + 1);
	int16_t *first = arena.take_n<int16_t>(nb + 1), *last = arena.take_n<int16_t>(nb + 1);
	uint8_t *bc = arena.take(nb * L + 16);
	for (size_t i = 0; i < nb; i++) memcpy(bc + i * L, cl[i].barcode.data(), L);
	const int nm = cfg.paired_end ? 2 : 1;
	Matrix mseq[2], mqual[2], mout[2];
	uint16_t *mlen[2] = {nullptr, nullptr}, *mlk[2] = {nullptr, nullptr};
	if (nb > 0 && fused) {
		size_t stride = 1;
		for (size_t i = 0; i < nb; i++) {
			stride = std::max(stride, std::max(cl[i].l2.n, cl[i].l4.n));
			if (cfg.paired_end) stride = std::max(stride, std::max(cl[i].m2[1].n, cl[i].m2[3].n));
		}
		if (stride > kMaxRow) { res.err = "Read longer than 65535 bases: not supported by this build."; res.nclusters = 0; return; }
		sk_fused_args fa;
		memset(&fa, 0, sizeof fa);
		fa.n = (int64_t)nb; fa.n_mates = nm; fa.stride = (int)stride; fa.min_baseq = cfg.q;
		for (int m = 0; m < nm; m++) {
			mseq[m].alloc(nb, stride, arena);
			mqual[m].alloc(nb, stride, arena);
			mlen[m] = arena.take_n<uint16_t>(nb);
			mlk[m] = arena.take_n<uint16_t>(nb);
			for (size_t i = 0; i < nb; i++) {
				const host::Line &sl = m ? cl[i].m2[1] : cl[i].l2, &ql = m ? cl[i].m2[3] : cl[i].l4;
				const size_t ls = strip_nl(sl), lq = strip_nl(ql);
				mseq[m].put(i, sl.p, ls);
				mqual[m].put(i, ql.p, lq);
				mlen[m][i] = (uint16_t)std::min(host::trim_end_len(ql.p, ql.n), lq);      // trim: n = qual.trim_end().len()
			}
			mout[m].alloc(nb, stride, arena);
			fa.mate[m].seq = mseq[m].data; fa.mate[m].qual = mqual[m].data; fa.mate[m].len = mlen[m];
			fa.mate[m].out_seq = cfg.do_mask ? mout[m].data : nullptr;
			fa.mate[m].lowest_k = cfg.do_trim ? mlk[m] : nullptr;
		}
		if (L > 0) {
			fa.bc = bc; fa.bc_stride = (int)L; fa.assign = assign;
			fa.lowest_diff = lowest; fa.first_idx = first; fa.last_idx = last;
		}
		host::GpuLease g;
		check(g.ctx(), sk_fused_pass(g.ctx(), &fa), "sk_fused_pass");
	} else if (nb > 0 && L > 0) {
		host::GpuLease g;
		check(g.ctx(), sk_demux_assign(g.ctx(), bc, (int)L, (int64_t)nb, assign, lowest, first, last), "sk_demux_assign");
	}
	// :190-194 — in a dry run the barcodes that matched no sample are counted: on the device, from the same matrix.  The
	// census table lives in ONE context (slot 0), whichever slot assigned the block.
	const bool dev_census = cfg.dry_run && L > 0 && L <= kCensusMaxLen;
	if (dev_census && nb > 0) {
		host::GpuLease g(0);
		check(g.ctx(), sk_census_add(g.ctx(), bc, (int)L, (int)L, (int64_t)nb, assign, (int64_t)base), "sk_census_add");
	}

	// body of one written record: verbatim lines, or the lines `mask by quality` then `trim by quality` would print
	auto emit_body = [&](std::string &w, const host::Line &sl, const host::Line &pl, const host::Line &ql, size_t i, int m) -> bool {
		if (!fused) { w.append(sl.p, sl.n); w.append(pl.p, pl.n); w.append(ql.p, ql.n); return true; }
		const size_t ls = strip_nl(sl), lq = strip_nl(ql);
		const char *seq_p = sl.p, *qual_p = ql.p;
		size_t seq_n = sl.n;                                                // current seq line incl. its newline (trim-only view)
		if (cfg.do_mask) {                                                  // src/fasta_mask_by_quality.rs:32-45
			if (ls != lq) { res.err = "Read sequence and base qualities are of different length."; return false; }
			if (!host::is_ascii(sl.p, sl.n) || !host::is_ascii(ql.p, ql.n)) { res.err = "Non-ASCII read lines are not supported together with --mask-by-quality."; return false; }
			seq_p = reinterpret_cast<const char *>(mout[m].data) + i * (size_t)mout[m].stride;
			seq_n = ls + 1;                                                 // masked bases + the newline mask prints
		}
		if (cfg.do_trim) {                                                  // src/fasta_trim_by_quality.rs:44-48
			const size_t lk = mlk[m][i];
			if (lk == 0) { w.append("N\n+\n!\n", 6); return true; }
			if (lk > seq_n) { res.err = "byte index out of range of `seq`"; res.err_code = 101; return false; }
			w.append(seq_p, lk);                                            // &seq[..lowest_k] of the current seq line
			w.append("\n+\n", 3);
			w.append(qual_p, lk);
			w.push_back('\n');
		} else {
			w.append(seq_p, ls); w.push_back('\n');
			w.append("+\n", 2);
			w.append(qual_p, lq); w.push_back('\n');
		}
		return true;
	};

	// ---- emit in input order (src/fasta_demultiplex.rs:168-238) ------------------------------------------------------
	if (!cfg.dry_run && nb > 0) {
		// room for every sample's share of the block at once (grown append by append, a sample's string was copied over and over)
		std::vector<uint32_t> share((size_t)S, 0u);
		for (size_t i = 0; i < nb; i++) if (assign[i] >= 0) share[(size_t)assign[i]]++;
		const size_t per1 = blk[0].size() / nb + 32, per2 = cfg.paired_end ? blk[1].size() / nb + 32 : 0;
		for (int s = 0; s < S; s++) {
			if (!share[(size_t)s]) continue;
			res.out1[(size_t)s].reserve((size_t)share[(size_t)s] * per1 + 64);
			if (cfg.paired_end) res.out2[(size_t)s].reserve((size_t)share[(size_t)s] * per2 + 64);
		}
	}
	std::string umi;
	char wbuf[1024];
	for (size_t i = 0; i < nb; i++) {
		ClusterV &c = cl[i];
		bool write_read_out = false;
		if (assign[i] >= 0) {                                               // :173-179
			res.identified += 1;
			res.per_sample[assign[i]] += 1;
			write_read_out = !cfg.dry_run;
		} else if (assign[i] == SK_ASSIGN_AMBIGUOUS) {                      // :181-189
			const Sample &a = samples[first[i]], &b = samples[last[i]];
			snprintf(wbuf, sizeof wbuf, "WARNING: Sequenced barcode %s was an equally good match (%u mismatches) for samples %s (%s) and %s (%s), and was therefore not assigned to any sample.\n",
			         c.barcode.c_str(), (unsigned)lowest[i], a.name.c_str(), a.barcode.c_str(), b.name.c_str(), b.barcode.c_str());
			res.warn += wbuf;
		} else if (cfg.dry_run) {                                           // :190-194
			if (!dev_census || !census_keyable(c.barcode.data(), c.barcode.size())) res.extras.add(c.barcode, (int64_t)(base + i));
		}
		if (!write_read_out) continue;
		const Sample &sm = samples[assign[i]];
		umi.clear();                                                        // :200-203 (chars().zip(chars()))
		{
			const uint8_t *sp = reinterpret_cast<const uint8_t *>(sm.barcode.data()), *bp = reinterpret_cast<const uint8_t *>(c.barcode.data());
			size_t a = 0, b = 0;
			while (a < sm.barcode.size() && b < c.barcode.size()) {
				const size_t la = u8len(sp[a]), lb = u8len(bp[b]);
				if (la == 1 && sp[a] == 'U') umi.append(c.barcode, b, lb);
				a += la; b += lb;
			}
		}
		// header with the BC field drained (:145), then trim_end (:206)
		auto emit_header = [&](std::string &w, const host::Line &h, bool drain, size_t st, size_t en) {
			const size_t before = w.size();
			if (drain) { w.append(h.p, st); w.append(h.p + en, h.n - en); }
			else w.append(h.p, h.n);
			w.resize(before + host::trim_end_len(w.data() + before, w.size() - before));
			if (!umi.empty()) { w.append(" UMI:", 5); w.append(umi); }       // :207
			w.push_back('\n');                                              // :208
		};
		std::string &w1 = res.out1[assign[i]];
		emit_header(w1, c.h, cfg.nindex == 0, c.bc_st, c.bc_en);
		if (!emit_body(w1, c.l2, c.l3, c.l4, i, 0)) { res.nclusters = i; return; }           // :209-212
		if (cfg.paired_end) {                                               // :215-237
			std::string &w2 = res.out2[assign[i]];
			size_t st = 0, en = 0;
			bool drain = false;
			if (cfg.nindex == 0) drain = host::find_bc_field(std::string(c.m2[0].p, c.m2[0].n), st, en);
			emit_header(w2, c.m2[0], drain, st, en);
			if (!emit_body(w2, c.m2[1], c.m2[2], c.m2[3], i, 1)) { res.nclusters = i; return; }
		}
	}
}

static const size_t kDemuxBlockRecords = 1u << 15;
static const size_t kCutsAhead = 32;                     // blocks the reader thread may be ahead of the workers (a block of 32 k records of 150 bases is 11 MB)

static int demultiplex(int argc, char **argv)
{
	// --mask-by-quality=Q / --trim-by-quality=Q are extensions of this build (not in the reference's usage text): every
	// written record additionally goes through `fasta mask by quality Q` and/or `fasta trim by quality Q`, computed in
	// the same device pass as the barcode assignment (sk_fused_pass) instead of by piping each sample file through them.
	std::vector<host::Opt> opts = {{"--parallel", false, false, ""}, {"--index1", true, false, ""}, {"--index2", true, false, ""}, {"--dry-run", true, false, ""},
	                               {"--mask-by-quality", true, false, ""}, {"--trim-by-quality", true, false, ""}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 3) || pos.size() < 2) error("Invalid arguments.\n%s", USAGE_DEMUX);
	const bool do_mask = opts[4].present, do_trim = opts[5].present;
	const uint8_t mask_q = do_mask ? parse_min_baseq(opts[4].value) : 0, trim_q = do_trim ? parse_min_baseq(opts[5].value) : 0;
	if (do_mask && do_trim && mask_q != trim_q) error("--mask-by-quality and --trim-by-quality must use the same threshold in one pass.");
	uint64_t dry_run = 0;                                                   // :33-36
	if (!host::parse_uint(opts[3].value.c_str(), UINT64_MAX, dry_run)) dry_run = 0;
	if (dry_run == 0 && !opts[3].value.empty()) error("In --dry-run=N, N must be 64-bit positive integer.");

	// :41-55 — FileReader::new for the reads and the index files (open errors come before the sheet is read)
	std::vector<std::unique_ptr<host::RecordBlocks>> files(4);
	{ host::LineReader probe(pos[1]); }
	files[0].reset(new host::RecordBlocks(pos[1], 4));
	const bool paired_end = pos.size() == 3 && !pos[2].empty();
	if (paired_end) { { host::LineReader probe(pos[2]); } files[1].reset(new host::RecordBlocks(pos[2], 4)); }
	int nindex = 0;
	for (int k = 1; k <= 2; k++)
		if (!opts[k].value.empty()) { { host::LineReader probe(opts[k].value); } files[2 + nindex].reset(new host::RecordBlocks(opts[k].value, 4)); nindex++; }

	host::gpu_warmup();
	fputs("Reading sample sheet...\n", stderr);                             // :58
	std::vector<Sample> samples;
	g_samples = &samples;
	host::at_exit_flush(close_outputs);
	size_t barcode_len = 0;
	{
		host::LineReader sheet(pos[0]);
		std::string line;
		for (;;) {
			const bool ok = sheet.read_line(line);                          // :63
			if (sheet.bad_utf8()) { close_outputs(); error("I/O error while reading from file."); }
			if (!ok) break;
			if (line[0] == '#') continue;                                   // :64
			const size_t off = host::trim_start_off(line), end = host::trim_end_len(line);      // :65 line.trim().split('\t')
			const std::string t = line.substr(off, end > off ? end - off : 0);
			const size_t tab1 = t.find('\t');
			if (tab1 == std::string::npos) continue;                        // :66
			const size_t tab2 = t.find('\t', tab1 + 1);
			Sample sm;
			sm.name = t.substr(0, tab1);
			sm.barcode = t.substr(tab1 + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab1 - 1);
			if (sm.barcode.empty()) error("Sample %s has no barcode.", sm.name.c_str());                 // :68
			if (barcode_len == 0) barcode_len = sm.barcode.size();          // :69-70
			else if (sm.barcode.size() != barcode_len) error("Barcodes in sample sheet must all be of same length.");   // :71-73
			if (dry_run > 0) {
			} else if (paired_end) {                                        // :79-83
				host::gz_deflate_on_device(true);                        // (this command has the GPU anyway: the per-sample members are deflated there)
				sm.out[0].reset(new host::GzWriter(sm.name + "_1.fq.gz"));
				sm.out[1].reset(new host::GzWriter(sm.name + "_2.fq.gz"));
			} else {                                                        // :84-87
				host::gz_deflate_on_device(true);
				sm.out[0].reset(new host::GzWriter(sm.name + ".fq.gz"));
			}
			samples.push_back(std::move(sm));
		}
	}
	for (size_t s = 0; s < samples.size(); s++)                             // :98-104
		for (size_t k = s + 1; k < samples.size(); k++)
			if (samples[s].name == samples[k].name) error("Sample %s is listed multiple times in sample sheet.", samples[s].name.c_str());

	const int S = (int)samples.size();
	if (S > SK_MAX_SAMPLES) error("Sample sheet has more than %d samples: not supported by this build.", SK_MAX_SAMPLES);
	if (barcode_len > SK_MAX_BARCODE_LEN) error("Barcodes longer than %d bytes are not supported by this build.", SK_MAX_BARCODE_LEN);
	{
		std::vector<uint8_t> table((size_t)S * barcode_len);
		for (int s = 0; s < S; s++) memcpy(table.data() + (size_t)s * barcode_len, samples[s].barcode.data(), barcode_len);
		// best / equally_fine / lowest_diff are read for reads that matched something (the warning, :184-188) and never for
		// the others (:190-194): SK_DETAIL_MATCHED, which lets the plain command take the sheet's lookup table
		host::gpu_for_each([&](sk_ctx *c) {
			check(c, sk_set_barcodes(c, table.data(), S, (int)barcode_len, 1 /* MAX_BARCODE_DIFFERENCE :168 */), "sk_set_barcodes");
			check(c, sk_set_detail_mode(c, SK_DETAIL_MATCHED), "sk_set_detail_mode");
		});
	}

	fprintf(stderr, "Starting demultiplexing in %s end mode...\n", paired_end ? "paired" : "single");     // :106-107
	uint64_t total_reads = 0, identified_reads = 0;
	HostCensus extra_barcodes;                       // dry run: what the device census cannot key
	if (dry_run > 0) check(sk_census_reset(host::gpu()), "sk_census_reset");

	DemuxCfg cfg{&samples, barcode_len, paired_end, nindex, do_mask, do_trim, do_mask ? mask_q : trim_q, dry_run > 0};
	unsigned nthreads = host::cpu_budget();
	if (const char *e = getenv("SEQKIT_THREADS")) nthreads = (unsigned)atoi(e);
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 16) nthreads = 16;
	size_t block_records = kDemuxBlockRecords;
	if (const char *e = getenv("SEQKIT_BLOCK_RECORDS")) block_records = (size_t)atoll(e);     // tests use tiny blocks

	struct Pending { std::shared_ptr<std::vector<host::Bytes>> data; std::future<std::shared_ptr<DemuxOut>> fut; };
	std::deque<Pending> inflight;
	// SEQKIT_PROF=1: where the main thread's time went (stderr, at the end)
	const bool prof = getenv("SEQKIT_PROF") != nullptr;
	auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
	double t_read = 0, t_wait = 0, t_hand = 0, t_cut_wait = 0;
	const double t_loop0 = now();
	auto drain_one = [&]() {
		double t0 = now();
		std::shared_ptr<DemuxOut> r = inflight.front().fut.get();
		t_wait += now() - t0;
		t0 = now();
		struct Hand { double &acc, t0; decltype(now) &clk; ~Hand() { acc += clk() - t0; } } hand{t_hand, t0, now};
		inflight.pop_front();
		fputs(r->warn.c_str(), stderr);
		total_reads += r->nclusters;                                        // :169
		identified_reads += r->identified;
		for (int s = 0; s < S; s++) samples[s].total_reads += r->per_sample[s];
		if (!cfg.dry_run)
			for (int s = 0; s < S; s++) {
				if (!r->out1[s].empty()) samples[s].out[0]->write(std::move(r->out1[s]));
				if (paired_end && !r->out2[s].empty()) samples[s].out[1]->write(std::move(r->out2[s]));
			}
		for (auto &e : r->extras.ents) extra_barcodes.add(e.label, e.first, e.count);
		if (!r->err.empty()) {
			for (auto &p : inflight) p.fut.wait();
			close_outputs();
			if (r->err_code == 101) panic(r->err.c_str());
			error("%s", r->err.c_str());
		}
	};

	// A reader thread cuts the input files into blocks (all files at the same record count) a few blocks ahead; this
	// thread starts a worker per block and hands the results on in input order.
	struct Cut { std::shared_ptr<std::vector<host::Bytes>> data; size_t want; };
	std::deque<Cut> cuts;
	std::mutex cm;
	std::condition_variable cv_cut, cv_room;
	bool cut_done = false;
	std::thread reader([&]() {
		uint64_t ahead = 0;
		for (;;) {
			size_t want = block_records;
			if (dry_run > 0) {                                              // :248 — blocks are cut so that exactly N clusters are read
				if (ahead >= dry_run) break;
				want = (size_t)std::min<uint64_t>(want, dry_run - ahead);
			}
			auto data = std::make_shared<std::vector<host::Bytes>>(4);
			const double tr0 = now();
			if (!files[0]->next(want, (*data)[0])) break;                   // :117 — the loop is driven by fastq_1
			for (int f = 1; f < 4; f++) if (files[f]) files[f]->next(want, (*data)[f]);
			t_read += now() - tr0;
			ahead += want;                           // a short block is the last one
			{
				std::unique_lock<std::mutex> lk(cm);
				cv_room.wait(lk, [&] { return cuts.size() < kCutsAhead; });      // (the reader runs ahead while the device contexts come up: 0.25 s in which it cuts a third of an 8 M-read file)
				cuts.push_back({data, want});
			}
			cv_cut.notify_one();
		}
		{
			std::lock_guard<std::mutex> lk(cm);
			cut_done = true;
		}
		cv_cut.notify_one();
	});
	uint64_t submitted = 0;
	for (;;) {
		Cut c;
		{
			const double tw0 = now();
			std::unique_lock<std::mutex> lk(cm);
			cv_cut.wait(lk, [&] { return !cuts.empty() || cut_done; });
			t_cut_wait += now() - tw0;
			if (cuts.empty()) break;
			c = std::move(cuts.front());
			cuts.pop_front();
		}
		cv_room.notify_one();
		auto data = c.data;
		Pending pd;
		pd.data = data;
		const uint64_t base = submitted;
		submitted += c.want;
		pd.fut = std::async(std::launch::async, [data, base, &cfg]() {
			auto res = std::make_shared<DemuxOut>();
			demux_block(cfg, data->data(), base, *res);
			return res;
		});
		inflight.push_back(std::move(pd));
		while (inflight.size() >= nthreads) drain_one();
	}
	reader.join();
	while (!inflight.empty()) drain_one();
	const double t_loop1 = now();

	// The blocks were dealt to several contexts (SEQKIT_GPUS devices x SEQKIT_CTXS_PER_GPU): their additive counters —
	// the reference's total_reads / identified_reads / sample.total_reads, :108-109,169,177-178 — are summed over all of
	// them (on one device by a kernel, across devices by RCCL), and must be what the ordered hand-off counted.
	if (S > 0 && barcode_len > 0) {
		std::vector<sk_ctx *> all;
		for (size_t k = 0; k < host::gpu_slots(); k++) all.push_back(host::gpu_slot(k));
		check(all[0], sk_counts_allreduce(all.data(), (int)all.size()), "sk_counts_allreduce");
		std::vector<uint64_t> dc((size_t)S + 3);
		check(all[0], sk_counts_get(all[0], dc.data()), "sk_counts_get");
		bool same = dc[(size_t)S] == total_reads && dc[(size_t)S + 1] == identified_reads;
		for (int s = 0; s < S && same; s++) same = dc[(size_t)s] == samples[s].total_reads;
		if (!same) { close_outputs(); error("internal: the device counters (%llu clusters, %llu identified) differ from the records handed on (%llu, %llu).",
		                                    (unsigned long long)dc[(size_t)S], (unsigned long long)dc[(size_t)S + 1], (unsigned long long)total_reads, (unsigned long long)identified_reads); }
	}

	if (dry_run > 0) {                                                      // :251-261
		fprintf(stderr, "Dry run completed with %llu clusters. Barcodes found:\n", (unsigned long long)total_reads);
		std::vector<CensusEnt> ents;                                        // :254-255 samples first, then the extras
		for (size_t k = 0; k < samples.size(); k++) ents.push_back({samples[k].name, samples[k].total_reads, (int64_t)k - (int64_t)samples.size()});
		std::vector<CensusEnt> extras = census_fetch(extra_barcodes, 100);
		ents.insert(ents.end(), extras.begin(), extras.end());
		print_most_frequent(ents);
	}
	close_outputs();
	host::out().flush();
	if (prof) fprintf(stderr, "demultiplex: %.3f s of the process before the block loop, block loop %.3f s (reader thread: read and cut %.3f; main thread: wait for a cut block %.3f, wait for the oldest "
	                          "result %.3f, hand it on %.3f), finishing the outputs %.3f s\n", t_loop0 - host::process_start_s(), t_loop1 - t_loop0, t_read, t_cut_wait, t_wait, t_hand, now() - t_loop1);
	fprintf(stderr, "%llu / %llu (%s%%) clusters carried a barcode matching one of the provided samples.\n",      // :263-264
	        (unsigned long long)identified_reads, (unsigned long long)total_reads,
	        host::fmt_pct((double)identified_reads / (double)total_reads * 100.0).c_str());
	return 0;
}

bool fasta_text_command(int argc, char **argv, bool before_trim_by_quality, int &rc);      // fasta_text.cpp

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
	// the order of src/fasta_main.rs:45-81 ("trim by quality" is tested before "trim")
	if (fasta_text_command(argc, argv, true, rc)) {}
	else if (argc >= 4 && is(1, "trim") && is(2, "by") && is(3, "quality")) rc = trim_by_quality(argc, argv);
	else if (argc >= 2 && is(1, "trim")) fasta_text_command(argc, argv, false, rc);
	else if (argc >= 4 && is(1, "mask") && is(2, "by") && is(3, "quality")) rc = mask_by_quality(argc, argv);
	else if (argc >= 3 && is(1, "gc") && is(2, "content")) rc = gc_content(argc, argv);
	else if (argc >= 3 && is(1, "add") && is(2, "barcode")) rc = add_barcode(argc, argv);
	else if (fasta_text_command(argc, argv, false, rc)) {}
	else if (argc >= 2 && is(1, "demultiplex")) rc = demultiplex(argc, argv);
	else if (argc >= 2 && is(1, "statistics")) rc = statistics(argc, argv);
	else fprintf(stderr, "%s\n", USAGE_TOP);
	host::out().flush();
	// everything is written and closed: what is left is taking the process apart (static destructors, the HIP runtime's exit handlers,
	// gigabytes of heap) — 0.16 s of a demultiplex of 8 M reads.  That is left to the kernel, as on the error path (host::error);
	// SEQKIT_SLOW_EXIT=1 (and SEQKIT_PROF, whose last lines are printed by destructors) returns from main instead.
	if (!getenv("SEQKIT_SLOW_EXIT") && !getenv("SEQKIT_PROF")) { host::flush_for_exit(); fflush(stdout); fflush(stderr); _exit(rc); }
	return rc;
}
