// fasta_text.cpp — the per-read TEXT commands of the reference's `fasta` binary.  There is no
// arithmetic in them worth a device: they are line filters, restated here so that the drop-in binary covers the same
// command surface.  Each function cites the reference lines it follows; output is what `print!` would have produced,
// byte for byte, including the places where the reference slices a String by byte offsets (a Rust panic, exit status
// 101, when the offset is out of range or inside a multi-byte character).
#include <cstring>
#include <string>
#include <vector>

#include "host_common.h"

using host::error;
using host::panic;

namespace {

// FileReader::read_line (src/common.rs:106-112): false at end of file, invalid UTF-8 is an I/O error
bool read_line(host::LineReader &f, std::string &line)
{
	const bool ok = f.read_line(line);
	if (f.bad_utf8()) error("I/O error while reading from file.");
	return ok;
}

inline bool starts_with(const std::string &s, char c) { return !s.empty() && s[0] == c; }
inline void put(const std::string &s) { host::out().write(s); }
inline void put(const char *p, size_t n) { host::out().write(p, n); }
inline void put(const char *lit) { host::out().write(lit, strlen(lit)); }

bool boundary(const std::string &s, size_t i) { return i == s.size() || (i < s.size() && ((uint8_t)s[i] & 0xC0) != 0x80); }

// &s[a..b]: panics like Rust's str indexing.  The arguments of a print! are evaluated before anything is written, so
// callers check every slice of one print! first.
void check_slice(const std::string &s, size_t a, size_t b)
{
	if (a > b) panic("slice index starts after its end");
	if (b > s.size()) panic("byte index out of range of string slice");
	if (!boundary(s, a) || !boundary(s, b)) panic("byte index is not a char boundary");
}
void put_slice(const std::string &s, size_t a, size_t b)
{
	check_slice(s, a, b);
	put(s.data() + a, b - a);
}

bool parse_usize(const std::string &s, uint64_t &out) { return host::parse_uint(s.c_str(), UINT64_MAX, out); }

// ---- fasta trim [--first=N] [--last=N] <fastq_file>                               src/fasta_trim.rs:14-48 ----
const char *USAGE_TRIM =
	"\nUsage:\n  fasta trim [options] <fastq_file>\n\nOptions:\n"
	"  --first=N          Remove first N bases of each read [default: 0].\n"
	"  --last=N           Remove last N bases of each read [default: 0].\n";

int trim(int argc, char **argv)
{
	std::vector<host::Opt> opts = {{"--first", true, false, "0"}, {"--last", true, false, "0"}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_TRIM);
	host::LineReader fasta_file(pos[0]);                                                    // :15
	uint64_t remove_first, remove_last;
	if (!parse_usize(opts[0].value, remove_first)) error("N must be a non-negative integer in --first=N.");     // :16-17
	if (!parse_usize(opts[1].value, remove_last)) error("N must be a non-negative integer in --last=N.");       // :18-19
	std::string line, seq, qual;
	while (read_line(fasta_file, line)) {                                                   // :24
		if (!starts_with(line, '>') && !starts_with(line, '@')) error("Invalid FASTA/FASTQ format encountered.");   // :25-27
		read_line(fasta_file, seq);                                                         // :29
		const uint64_t seq_len = host::trim_end_len(seq);                                   // :30
		const bool keep = remove_first + remove_last < seq_len;                             // :31 (usize addition wraps in a release build)
		if (keep) check_slice(seq, remove_first, seq_len - remove_last);
		put(line);
		if (keep) put_slice(seq, remove_first, seq_len - remove_last);                      // :32
		put("\n");
		if (starts_with(line, '@')) {                                                       // :37
			read_line(fasta_file, line);
			read_line(fasta_file, qual);
			if (keep) check_slice(qual, remove_first, seq_len - remove_last);
			put("+\n");
			if (keep) put_slice(qual, remove_first, seq_len - remove_last);                 // :41
			put("\n");
		}
	}
	return 0;
}

// ---- fasta extract dual umi [--first-bases=N] <interleaved_fastq>       src/fasta_extract_dual_umi.rs:14-72 ----
const char *USAGE_UMI =
	"\nUsage:\n  fasta extract dual umi [options] <interleaved_fastq>\n\nOptions:\n"
	"  --first-bases=N   First N bases of read contain UMI bases [default: 0]\n";

int extract_dual_umi(int argc, char **argv)
{
	std::vector<host::Opt> opts = {{"--first-bases", true, false, "0"}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_UMI);
	host::LineReader fastq(pos[0]);                                                         // :16
	uint64_t first_bases;
	if (!parse_usize(opts[0].value, first_bases)) error("N must be a non-negative integer in --first-bases=N.");   // :17-19
	std::string header_1, header_2, seq_1, seq_2, qual_1, qual_2, line;
	while (read_line(fastq, header_1)) {                                                    // :30
		bool fastq_format;
		if (starts_with(header_1, '@')) fastq_format = true;                                // :33-35
		else if (starts_with(header_1, '>')) fastq_format = false;
		else error("Header is not valid FASTA/FASTQ:\n%s", header_1.c_str());
		if (fastq_format) {                                                                 // :37-47
			read_line(fastq, seq_1); read_line(fastq, line); read_line(fastq, qual_1);
			read_line(fastq, header_2); read_line(fastq, seq_2); read_line(fastq, line); read_line(fastq, qual_2);
			if (!starts_with(header_2, '@')) error("Invalid FASTQ record found in input file.");
		} else {                                                                            // :48-55
			read_line(fastq, seq_1); read_line(fastq, header_2); read_line(fastq, seq_2);
			if (!starts_with(header_2, '>')) error("Invalid FASTA record found in input file.");
		}
		// :57-59  umi = seq_1[0..n] + "+" + seq_2[0..n]; the slices panic before anything of this pair is printed
		for (const std::string *s : {&seq_1, &seq_2}) {
			if (first_bases > s->size()) panic("byte index out of range of string slice");
			if (!boundary(*s, first_bases)) panic("byte index is not a char boundary");
		}
		const std::string umi = seq_1.substr(0, first_bases) + "+" + seq_2.substr(0, first_bases);
		auto emit = [&](const std::string &header, const std::string &seq, const std::string &qual) {
			put(header.data(), host::trim_end_len(header));
			put(" RX:"); put(umi); put("\n");
			put_slice(seq, first_bases, seq.size());
			if (fastq_format) { put("+\n"); put_slice(qual, first_bases, qual.size()); }
		};
		// :61-70 — every argument of the print! is evaluated before anything is written
		if (fastq_format)
			for (const std::string *q : {&qual_1, &qual_2}) {
				if (first_bases > q->size()) panic("byte index out of range of string slice");
				if (!boundary(*q, first_bases)) panic("byte index is not a char boundary");
			}
		emit(header_1, seq_1, qual_1);
		emit(header_2, seq_2, qual_2);
	}
	return 0;
}

// ---- fasta convert basespace <fastq_file>                                src/fasta_convert_basespace.rs:17-47 ----
const char *USAGE_BASESPACE =
	"\nUsage:\n  fasta convert basespace <fastq_file>\n\nDescription:\n"
	"FASTQ files from Illumina Basespace typically display adapter barcodes at\n"
	"the end of the FASTQ header, and have read identifiers that end in /1 or /2.\n"
	"This tool replaces the read identifiers by simple consecutive integers, and\n"
	"places a \"BC:\" prefix in front of the barcode. An example FASTQ header in\n"
	"the output could look like this: @412435 BC:TAGCTACT\n";

int convert_basespace(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 3, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_BASESPACE);
	host::LineReader fastq(pos[0]);                                                         // :19
	uint64_t num_read_pairs = 0;
	std::string header, line;
	char buf[32];
	while (read_line(fastq, header)) {                                                      // :25
		num_read_pairs += 1;
		snprintf(buf, sizeof buf, "@%llu", (unsigned long long)num_read_pairs);             // :27
		put(buf);
		const size_t end = host::trim_end_len(header);                                      // :32 header.trim_end().split(':').last()
		const size_t colon = header.rfind(':', end ? end - 1 : 0);
		const size_t from = (end && colon != std::string::npos && colon < end) ? colon + 1 : 0;
		if (end > from) { put(" BC:"); put(header.data() + from, end - from); }             // :33
		put("\n");                                                                          // :34
		if (starts_with(header, '@')) {                                                     // :36-39
			for (int k = 0; k < 3; k++) { read_line(fastq, line); put(line); }
		} else if (starts_with(header, '>')) {                                              // :40-41
			read_line(fastq, line); put(line);
		} else {
			error("Invalid FASTQ line:\n%s", header.c_str());                               // :42-44
		}
	}
	return 0;
}

// ---- fasta simplify read ids [--alphanumeric] [--discard-umi] <fastq_file>   src/fasta_simplify_read_ids.rs:19-62 ----
const char *USAGE_SIMPLIFY =
	"\nUsage:\n  fasta simplify read ids [options] <fastq_file>\n\nOptions:\n"
	"  --alphanumeric     Use letters a-z, A-Z and 0-9 in read identifiers\n"
	"  --discard-umi      Remove \"UMI:\" tags from read identifiers, if present\n";

int simplify_read_ids(int argc, char **argv)
{
	std::vector<host::Opt> opts = {{"--alphanumeric", false, false, ""}, {"--discard-umi", false, false, ""}};
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_SIMPLIFY);
	host::LineReader fasta_file(pos[0]);                                                    // :21
	const bool discard_umi = opts[1].present;                                               // :23 (--alphanumeric is parsed and unused, :22)
	uint64_t read_num = 0;
	std::string line;
	char buf[32];
	while (read_line(fasta_file, line)) {                                                   // :30
		const char prefix = line[0];                                                        // :33 first char; only '@' and '>' pass
		if (prefix != '@' && prefix != '>') error("Invalid FASTA/FASTQ format encountered.");          // :34-36
		read_num += 1;
		snprintf(buf, sizeof buf, "%c%llu", prefix, (unsigned long long)read_num);          // :39
		put(buf);
		size_t st, en;
		if (!discard_umi && host::find_umi_field(line, st, en)) put(line.data() + st, en - st);         // :42-46
		put("\n");                                                                          // :47
		read_line(fasta_file, line); put(line);                                             // :50-51
		if (prefix == '@') {                                                                // :54-59
			read_line(fasta_file, line);
			put("+\n");
			read_line(fasta_file, line); put(line);
		}
	}
	return 0;
}

// ---- fasta interleave <fastq_1> <fastq_2>                                        src/fasta_interleave.rs:9-35 ----
const char *USAGE_INTERLEAVE = "\nUsage:\n  fasta interleave <fastq_1> <fastq_2>\n";

int interleave(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_INTERLEAVE);
	host::LineReader fastq_1(pos[0]), fastq_2(pos[1]);                                      // :11-12
	std::string line;
	while (read_line(fastq_1, line)) {                                                      // :15
		int lines;
		if (starts_with(line, '@')) lines = 4;                                              // :16-18
		else if (starts_with(line, '>')) lines = 2;
		else error("Line is not FASTA/FASTQ format: %s", line.c_str());
		put(line);                                                                          // :19
		for (int k = 0; k < lines - 1; k++) { read_line(fastq_1, line); put(line); }        // :20-22
		read_line(fastq_2, line);                                                           // :24
		if ((lines == 4 && !starts_with(line, '@')) || (lines == 2 && !starts_with(line, '>')))        // :25-28
			error("Input files do not share a consistent format.");
		put(line);                                                                          // :29
		for (int k = 0; k < lines - 1; k++) { read_line(fastq_2, line); put(line); }        // :30-32
	}
	return 0;
}

// ---- fasta check <fasta/fastq>                                                    src/fasta_check.rs:14-70 ----
const char *USAGE_CHECK =
	"\nUsage:\n  fasta check <fasta/fastq>\n\nDescription:\n"
	"Checks that the input FASTA or FASTQ file is correctly formatted, and reports\n"
	"the line number if any malformatted lines are found.\n";

struct ReaderWithMemory {                                  // :14-45
	host::LineReader file;
	uint64_t lines_read = 0;
	std::vector<std::string> prev_lines;                   // the last 10 lines
	explicit ReaderWithMemory(const std::string &path) : file(path) {}
	bool next(std::string &line)
	{
		if (!read_line(file, line)) return false;
		prev_lines.push_back(line);
		if (prev_lines.size() > 10) prev_lines.erase(prev_lines.begin());
		lines_read += 1;
		return true;
	}
	std::string history() const
	{
		std::string h;
		for (const std::string &l : prev_lines) { h += l; h += "\n"; }
		return h;
	}
};

int check(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_CHECK);
	ReaderWithMemory fasta(pos[0]);                                                         // :50
	std::string line;
	while (fasta.next(line)) {                                                              // :53
		if (starts_with(line, '>')) {
			fasta.next(line);
		} else if (starts_with(line, '@')) {
			fasta.next(line);
			fasta.next(line);
			if (!starts_with(line, '+'))                                                    // :59-62
				error("Missing quality header prefix '+' on line %llu:\n%s\n", (unsigned long long)fasta.lines_read, fasta.history().c_str());
			fasta.next(line);
		} else {                                                                            // :64-67
			error("Missing header prefix '>' or '@' on line %llu:\n%s\n", (unsigned long long)fasta.lines_read, fasta.history().c_str());
		}
	}
	return 0;
}

// ---- fasta to raw <fasta_file>                                                   src/fasta_to_raw.rs:9-29 ----
const char *USAGE_TO_RAW = "\nUsage:\n  fasta to raw <fasta_file>\n";

int to_raw(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 3, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_TO_RAW);
	host::LineReader fasta_file(pos[0]);
	std::string line;
	while (read_line(fasta_file, line)) {
		if (starts_with(line, '>')) {                                                       // :15-17
			read_line(fasta_file, line); put(line);
		} else if (starts_with(line, '@')) {                                                // :18-23
			read_line(fasta_file, line); put(line);
			read_line(fasta_file, line);
			read_line(fasta_file, line);
		} else {
			error("Invalid FASTA/FASTQ format encountered.");                               // :25
		}
	}
	return 0;
}

// ---- fasta add base qualities <fasta> <baseq>                        src/fasta_add_base_qualities.rs:12-31 ----
const char *USAGE_ADD_BASEQ =
	"\nUsage:\n  fasta add base qualities <fasta> <baseq>\n\n"
	"Converts a FASTA file into a FASTQ file based on user-specified dummy base\nquality values.\n";

int add_base_qualities(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_ADD_BASEQ);
	host::LineReader fasta_file(pos[0]);                                                    // :14
	uint64_t baseq;
	if (!host::parse_uint(pos[1].c_str(), 255, baseq)) error("Base quality must be between 0 - 255.");       // :15-16
	const uint8_t q = (uint8_t)(33 + baseq);                                                // :26 u8 addition wraps in a release build
	std::string line;
	while (read_line(fasta_file, line)) {
		if (!starts_with(line, '>')) error("Invalid FASTA format encountered.");            // :27-29
		put("@"); put(line.data() + 1, line.size() - 1);                                    // :21
		read_line(fasta_file, line);                                                        // :22
		put(line);                                                                          // :24
		if (line.empty()) panic("capacity overflow");                                       // :23 line.len() - 1 wraps to usize::MAX
		const size_t seq_len = line.size() - 1;
		if (q >= 0x80 && seq_len > 0) panic("called `Result::unwrap()` on an `Err` value: Utf8Error");       // :25 from_utf8(..).unwrap()
		put("+\n"); put(std::string(seq_len, (char)q)); put("\n");
	}
	return 0;
}

// ---- fasta remove base qualities <fastq_file>                     src/fasta_remove_base_qualities.rs:9-27 ----
const char *USAGE_REMOVE_BASEQ = "\nUsage:\n  fasta remove base qualities <fastq_file>\n";

int remove_base_qualities(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 1) || pos.size() != 1) error("Invalid arguments.\n%s", USAGE_REMOVE_BASEQ);
	host::LineReader fastq_file(pos[0]);
	std::string line;
	while (read_line(fastq_file, line)) {
		if (!starts_with(line, '@')) error("Invalid FASTQ format encountered.");            // :23-25
		put(">"); put(line.data() + 1, line.size() - 1);                                    // :16
		read_line(fastq_file, line); put(line);                                             // :17-18
		read_line(fastq_file, line);                                                        // :20-21
		read_line(fastq_file, line);
	}
	return 0;
}

// ---- fasta deinterleave <interleaved_fastq> <out_prefix>                      src/fasta_deinterleave.rs:9-39 ----
const char *USAGE_DEINTERLEAVE = "\nUsage:\n  fasta deinterleave <interleaved_fastq> <out_prefix>\n";

host::GzWriter *g_deint[2] = {nullptr, nullptr};
void close_deint() { for (auto *w : g_deint) if (w) w->close(); }

int deinterleave(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 2, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_DEINTERLEAVE);
	host::LineReader fastq(pos[0]);                                                         // :11
	host::GzWriter out_1(pos[1] + "_1.fq.gz"), out_2(pos[1] + "_2.fq.gz");                  // :13-16
	g_deint[0] = &out_1; g_deint[1] = &out_2;
	host::at_exit_flush(close_deint);
	std::string line;
	while (read_line(fastq, line)) {                                                        // :19
		int lines;
		if (starts_with(line, '@')) lines = 4;                                              // :20-22
		else if (starts_with(line, '>')) lines = 2;
		else error("Line is not FASTA/FASTQ format: %s", line.c_str());
		out_1.write(line);                                                                  // :23
		for (int k = 0; k < lines - 1; k++) { read_line(fastq, line); out_1.write(line); }  // :24-26
		read_line(fastq, line);                                                             // :28
		if ((lines == 4 && !starts_with(line, '@')) || (lines == 2 && !starts_with(line, '>')))          // :29-32
			error("Interleaved FASTA records are not in consistent format.");
		out_2.write(line);                                                                  // :33
		for (int k = 0; k < lines - 1; k++) { read_line(fastq, line); out_2.write(line); }  // :34-36
	}
	close_deint();
	g_deint[0] = g_deint[1] = nullptr;
	return 0;
}

// ---- fasta split into anchors <fastq> <anchor_len>                       src/fasta_split_into_anchors.rs:10-45 ----
const char *USAGE_ANCHORS = "\nUsage:\n  fasta split into anchors <fastq> <anchor_len>\n";

int split_into_anchors(int argc, char **argv)
{
	std::vector<host::Opt> opts;
	std::vector<std::string> pos;
	if (!host::parse_args(argc, argv, 4, opts, pos, 2) || pos.size() != 2) error("Invalid arguments.\n%s", USAGE_ANCHORS);
	host::LineReader fastq(pos[0]);                                                         // :12
	uint64_t anchor_len;
	if (!parse_usize(pos[1], anchor_len)) error("<anchor_len> must be a positive integer.");                 // :13-14
	uint64_t reads = 0;
	std::string header, seq, qual, line;
	char buf[48];
	while (read_line(fastq, header)) {                                                      // :22
		reads += 1;
		read_line(fastq, seq);                                                              // :25
		const uint64_t seq_len = host::trim_end_len(seq);                                   // :26
		// :27 — a FASTQ record that is too short is skipped WITHOUT reading its '+' and quality lines: the reference
		// then takes those for the next record's header and bases.  (anchor_len * 2 wraps in a release build.)
		if (seq_len < anchor_len * 2) continue;
		if (starts_with(header, '@')) {                                                     // :29-36
			read_line(fastq, line);
			read_line(fastq, qual);
			check_slice(seq, 0, anchor_len); check_slice(qual, 0, anchor_len);
			snprintf(buf, sizeof buf, "@%llu\n", (unsigned long long)reads);
			put(buf); put_slice(seq, 0, anchor_len); put("\n+\n"); put_slice(qual, 0, anchor_len); put("\n");
			check_slice(seq, seq_len - anchor_len, seq_len); check_slice(qual, seq_len - anchor_len, seq_len);
			put(buf); put_slice(seq, seq_len - anchor_len, seq_len); put("\n+\n"); put_slice(qual, seq_len - anchor_len, seq_len); put("\n");
		} else if (starts_with(header, '>')) {                                              // :37-39
			check_slice(seq, 0, anchor_len);
			snprintf(buf, sizeof buf, ">%llu\n", (unsigned long long)reads);
			put(buf); put_slice(seq, 0, anchor_len); put("\n");
			check_slice(seq, seq_len - anchor_len, seq_len);
			put(buf); put_slice(seq, seq_len - anchor_len, seq_len); put("\n");
		} else {
			error("Header is not valid FASTA/FASTQ:\n%s", header.c_str());                  // :40-42
		}
	}
	return 0;
}

}  // namespace

// dispatch in the order of src/fasta_main.rs:45-81; returns false when argv names none of these commands
bool fasta_text_command(int argc, char **argv, bool before_trim_by_quality, int &rc)
{
	auto is = [&](int i, const char *w) { return argc > i && strcmp(argv[i], w) == 0; };
	if (before_trim_by_quality) {
		if (argc >= 2 && is(1, "check")) { rc = check(argc, argv); return true; }
		if (argc >= 3 && is(1, "to") && is(2, "raw")) { rc = to_raw(argc, argv); return true; }
		if (argc >= 4 && is(1, "add") && is(2, "base") && is(3, "qualities")) { rc = add_base_qualities(argc, argv); return true; }
		if (argc >= 4 && is(1, "remove") && is(2, "base") && is(3, "qualities")) { rc = remove_base_qualities(argc, argv); return true; }
		if (argc >= 4 && is(1, "simplify") && is(2, "read") && is(3, "ids")) { rc = simplify_read_ids(argc, argv); return true; }
		if (argc >= 2 && is(1, "interleave")) { rc = interleave(argc, argv); return true; }
		if (argc >= 2 && is(1, "deinterleave")) { rc = deinterleave(argc, argv); return true; }
		if (argc >= 4 && is(1, "split") && is(2, "into") && is(3, "anchors")) { rc = split_into_anchors(argc, argv); return true; }
		return false;
	}
	if (argc >= 2 && is(1, "trim")) { rc = trim(argc, argv); return true; }
	if (argc >= 4 && is(1, "extract") && is(2, "dual") && is(3, "umi")) { rc = extract_dual_umi(argc, argv); return true; }
	if (argc >= 3 && is(1, "convert") && is(2, "basespace")) { rc = convert_basespace(argc, argv); return true; }
	return false;
}
