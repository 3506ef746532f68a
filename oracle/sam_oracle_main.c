/*
 * sam_oracle — ORACLE restatement of the reference's `sam statistics` and
 * `sam fragment lengths` (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see seqkit_oracle.h).
 *
 *   sam statistics [--on-target=BED] <bam_file>                       src/sam_statistics.rs:14-116
 *   sam fragment lengths [--max-frag-size=F] [--reads=N] <bam_file>   src/sam_fragment_lengths.rs:14-48
 * dispatch: src/sam_main.rs:50-53.  BamReader: src/common.rs:121-157.
 *
 * The reference reads BAM through rust-htslib 0.31 (C htslib), which is not in the tree
 * and not in this image.  The record walk below follows the SAM/BAM specification
 * (SAMv1 §4.2): BGZF = concatenated gzip members; after the header each record is
 * block_size:u32 followed by a 32-byte fixed core with refID@0, flag@14, next_refID@20,
 * tlen@28.  --on-target (S2) is out of scope and is rejected.
 */
#include <zlib.h>

#include "cli_common.h"

static const char *USAGE_TOP = "\nUsage:\n  sam statistics <bam_file>\n  sam fragment lengths <bam_file>\n";
static const char *USAGE_STATS =
"\nUsage:\n  sam statistics [options] <bam_file>\n\nOptions:\n"
"  --on-target=BED   Count on-target% for regions in BED file [optional]\n";
static const char *USAGE_FRAG =
"\nUsage:\n  sam fragment lengths [options] <bam_file>\n\nOptions:\n"
"  --max-frag-size=F     Maximum fragment size [default: 5000]\n"
"  --reads=N             Finish after analyzing this many reads [default: Inf]\n";

/* ---- multi-member gzip (BGZF) byte stream ---------------------------------------- */
typedef struct {
	FILE *f;
	z_stream z;
	uint8_t in[1 << 16];
	int eof, started, raw;     /* raw: input is not gzip at all -> not a BAM */
} bgzf_t;

static void bgzf_open(bgzf_t *b, const char *path)
{
	memset(b, 0, sizeof(*b));
	if (strcmp(path, "-") == 0) b->f = stdin;
	else b->f = fopen(path, "rb");
	if (!b->f) {
		if (strcmp(path, "-") == 0) oc_error("Failed to read BAM file from standard input.");
		oc_error("Cannot open BAM file '%s'", path);
	}
	if (inflateInit2(&b->z, 15 + 16) != Z_OK) oc_error("Cannot open BAM file '%s'", path);
}

/* returns bytes produced (< n only at end of stream); -1 on a corrupt stream */
static long bgzf_read(bgzf_t *b, void *dst, size_t n)
{
	uint8_t *out = (uint8_t *)dst;
	size_t got = 0;
	while (got < n) {
		if (b->z.avail_in == 0 && !b->eof) {
			size_t r = fread(b->in, 1, sizeof b->in, b->f);
			if (r == 0) b->eof = 1;
			b->z.next_in = b->in;
			b->z.avail_in = (uInt)r;
		}
		if (b->z.avail_in == 0 && b->eof) break;
		b->z.next_out = out + got;
		b->z.avail_out = (uInt)(n - got);
		int rc = inflate(&b->z, Z_NO_FLUSH);
		got = n - b->z.avail_out;
		if (rc == Z_STREAM_END) { inflateReset(&b->z); continue; }
		if (rc != Z_OK && rc != Z_BUF_ERROR) return -1;
		if (rc == Z_BUF_ERROR && b->z.avail_in == 0 && b->eof) break;
	}
	return (long)got;
}

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static char **g_ref_names = NULL;
static uint32_t g_n_ref = 0;

static void bam_read_header(bgzf_t *b, const char *path)
{
	uint8_t h[8];
	if (bgzf_read(b, h, 8) != 8 || memcmp(h, "BAM\1", 4) != 0) oc_error("Cannot open BAM file '%s'", path);
	uint32_t l_text = le32(h + 4);
	uint8_t buf[4096];
	while (l_text) { size_t c = l_text > sizeof buf ? sizeof buf : l_text; if (bgzf_read(b, buf, c) != (long)c) oc_error("Cannot open BAM file '%s'", path); l_text -= (uint32_t)c; }
	if (bgzf_read(b, h, 4) != 4) oc_error("Cannot open BAM file '%s'", path);
	uint32_t n_ref = le32(h);
	g_n_ref = n_ref;
	g_ref_names = (char **)calloc(n_ref ? n_ref : 1, sizeof(char *));
	for (uint32_t i = 0; i < n_ref; i++) {
		if (bgzf_read(b, h, 4) != 4) oc_error("Cannot open BAM file '%s'", path);
		uint32_t l_name = le32(h);
		g_ref_names[i] = (char *)calloc(l_name + 1, 1);
		if (l_name && bgzf_read(b, g_ref_names[i], l_name) != (long)l_name) oc_error("Cannot open BAM file '%s'", path);
		if (bgzf_read(b, h, 4) != 4) oc_error("Cannot open BAM file '%s'", path);      /* l_ref */
	}
}

typedef struct { uint16_t flag; int32_t tid, mtid, tlen, pos; } core_t;

/* 1 = record, 0 = clean end of file; errors exit like src/common.rs:150-154 */
static int bam_next(bgzf_t *b, core_t *c)
{
	uint8_t h[4];
	long r = bgzf_read(b, h, 4);
	if (r == 0) return 0;
	if (r < 0) oc_error("Invalid BAM record.");
	if (r != 4) oc_error("BAM file ended prematurely.");
	uint32_t block_size = le32(h);
	if (block_size < 32) oc_error("Invalid BAM record.");
	uint8_t core[32];
	r = bgzf_read(b, core, 32);
	if (r < 0) oc_error("Invalid BAM record.");
	if (r != 32) oc_error("BAM file ended prematurely.");
	c->tid = (int32_t)le32(core + 0);
	c->pos = (int32_t)le32(core + 4);
	c->flag = le16(core + 14);
	c->mtid = (int32_t)le32(core + 20);
	c->tlen = (int32_t)le32(core + 28);
	uint32_t rest = block_size - 32;
	uint8_t buf[4096];
	while (rest) {
		size_t n = rest > sizeof buf ? sizeof buf : rest;
		r = bgzf_read(b, buf, n);
		if (r < 0) oc_error("Invalid BAM record.");
		if (r != (long)n) oc_error("BAM file ended prematurely.");
		rest -= (uint32_t)n;
	}
	return 1;
}

/* src/sam_statistics.rs:14-116 (without --on-target) */
static int statistics(int argc, char **argv)
{
	oc_opt opts[1] = {{"--on-target", 1, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 2, opts, 1, pos, &npos, 1) || npos != 1)
		oc_error("Invalid arguments.\n%s", USAGE_STATS);
	if (opts[0].value && opts[0].value[0]) {
		fputs("sam_oracle: --on-target is outside the oracle's scope (SURVEY.md §8a S2)\n", stderr);
		return 2;
	}
	bgzf_t *b = (bgzf_t *)malloc(sizeof(bgzf_t));
	bgzf_open(b, pos[0]);
	bam_read_header(b, pos[0]);
	uint64_t counters[3] = {0, 0, 0};
	core_t c;
	while (bam_next(b, &c))                                                    /* :63-69 */
		orc_bam_flag_tlen(&c.flag, &c.tid, &c.mtid, &c.tlen, 1, 0, counters, NULL, NULL);
	char p1[64], p2[64];                                                       /* :109-111 */
	oc_fmt_pct(p1, sizeof p1, (double)counters[1] / (double)counters[0] * 100.0);
	oc_fmt_pct(p2, sizeof p2, (double)counters[2] / (double)counters[1] * 100.0);
	printf("Total reads: %llu\n", (unsigned long long)counters[0]);
	printf("Aligned reads: %llu (%s%% of all reads)\n", (unsigned long long)counters[1], p1);
	printf("Duplicate reads: %llu (%s%% of aligned reads)\n", (unsigned long long)counters[2], p2);
	return 0;
}

/* src/sam_fragment_lengths.rs:14-48 */
static int fragment_lengths(int argc, char **argv)
{
	oc_opt opts[2] = {{"--max-frag-size", 1, NULL}, {"--reads", 1, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 3, opts, 2, pos, &npos, 1) || npos != 1)
		oc_error("Invalid arguments.\n%s", USAGE_FRAG);
	uint64_t max_frag = 5000, stop = UINT64_MAX;
	if (opts[0].value && !oc_parse_uint(opts[0].value, UINT64_MAX, &max_frag)) oc_panic("--max-frag-size parse().unwrap()");
	if (opts[1].value && strcmp(opts[1].value, "Inf") != 0 && !oc_parse_uint(opts[1].value, UINT64_MAX, &stop)) oc_panic("--reads parse().unwrap()");
	if (max_frag > 0x7fffffffULL) max_frag = 0x7fffffffULL;   /* |tlen| never exceeds 2^31 */
	uint64_t *hist = (uint64_t *)calloc(max_frag + 1, sizeof(uint64_t));
	if (!hist) oc_panic("capacity overflow");
	bgzf_t *b = (bgzf_t *)malloc(sizeof(bgzf_t));
	bgzf_open(b, pos[0]);
	bam_read_header(b, pos[0]);
	uint64_t total = 0;
	core_t c;
	while (bam_next(b, &c)) {                                                  /* :29-43 */
		uint64_t before = total;
		orc_bam_flag_tlen(&c.flag, &c.tid, &c.mtid, &c.tlen, 1, (int32_t)max_frag, NULL, hist, &total);
		if (total != before && total >= stop) break;                           /* :42 */
	}
	for (uint64_t size = 1; size < max_frag + 1; size++)                       /* :45-47 */
		printf("%llu\t%llu\n", (unsigned long long)size, (unsigned long long)hist[size]);
	return 0;
}

static const char *USAGE_FRAGMENTS =
"\nUsage:\n  sam fragments [options] <bam_file>\n\nOptions:\n"
"  --min-size=N     Minimum fragment size [default: 0]\n"
"  --max-size=N     Maximum fragment size [default: 5000]\n";

/* str::parse::<i64>(): optional sign, digits */
static int parse_i64(const char *s, int64_t *out)
{
	int neg = 0;
	if (*s == '+' || *s == '-') { neg = *s == '-'; s++; }
	uint64_t v;
	if (!oc_parse_uint(s, neg ? 9223372036854775808ULL : 9223372036854775807ULL, &v)) return 0;
	*out = neg ? (int64_t)(0 - v) : (int64_t)v;
	return 1;
}

/* src/sam_fragments.rs:14-43 */
static int fragments(int argc, char **argv)
{
	oc_opt opts[2] = {{"--min-size", 1, NULL}, {"--max-size", 1, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 2, opts, 2, pos, &npos, 1) || npos != 1)
		oc_error("Invalid arguments.\n%s", USAGE_FRAGMENTS);
	int64_t min_size = 0, max_size = 5000;
	if (opts[0].value && !parse_i64(opts[0].value, &min_size)) oc_panic("--min-size parse().unwrap()");
	if (opts[1].value && !parse_i64(opts[1].value, &max_size)) oc_panic("--max-size parse().unwrap()");
	bgzf_t *b = (bgzf_t *)malloc(sizeof(bgzf_t));
	bgzf_open(b, pos[0]);
	bam_read_header(b, pos[0]);
	core_t c;
	while (bam_next(b, &c)) {
		uint8_t keep;
		if (!orc_fragments_keep(&c.flag, &c.tid, &c.mtid, &c.tlen, 1, min_size, max_size, &keep)) continue;
		if (c.tid < 0 || (uint32_t)c.tid >= g_n_ref) oc_panic("index out of bounds: chr_names[tid]");
		int64_t t = c.tlen; if (t < 0) t = -t;
		printf("%s\t%lld\t%lld\n", g_ref_names[c.tid], (long long)c.pos, (long long)c.pos + (long long)t);   /* :41 */
	}
	return 0;
}

int main(int argc, char **argv)
{
	int rc;
	if (argc >= 2 && !strcmp(argv[1], "fragments")) rc = fragments(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "statistics")) rc = statistics(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "fragment") && !strcmp(argv[2], "lengths")) rc = fragment_lengths(argc, argv);
	else { fprintf(stderr, "%s\n", USAGE_TOP); rc = 0; }
	fflush(stdout);
	return rc;
}
