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

/* src/sam_main.rs:16-39, trailing blanks included */
static const char *USAGE_TOP =
"\nUsage:\n"
"  sam merge <bam_files>...\n"
"  sam consensus <bam_file>\n"
"  sam count <bam_file> <regions.bed>\n"
"  sam coverage histogram <bam_file>\n"
"  sam fragments <bam_file>\n"
"  sam fragment lengths <bam_file>\n"
"  sam mark duplicates <bam_file>\n"
"  sam minimize <bam_file>\n"
"  sam statistics <bam_file>\n"
"  sam subsample <bam_file> <fraction>  \n"
"  sam tags from qname <bam_file>\n"
"  sam qname from tags <bam_file>\n"
"  sam trim qnames <bam_file>  \n"
"\n"
"Extract reads from BAM files:  \n"
"  sam to fasta <bam_file> <out_prefix>\n"
"  sam to fastq <bam_file> <out_prefix>  \n"
"  sam to interleaved fasta <bam_file>\n"
"  sam to interleaved fastq <bam_file>\n"
"  sam to interleaved raw <bam_file>\n"
"  sam to raw <bam_file> <out_prefix>\n";
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

typedef struct { uint16_t flag; int32_t tid, mtid, tlen, pos, mpos; } core_t;

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
	c->mpos = (int32_t)le32(core + 24);
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

/* ---- f4: sam to raw|fasta|fastq (src/sam_to_fastq.rs:61-149) ------------------------------ */
/* whole record: the fixed core and everything after it (qname, cigar, seq, qual, aux) */
typedef struct { core_t c; uint8_t mapq; uint32_t l_read_name, n_cigar, l_seq; oc_str body; } record_t;

static int bam_next_full(bgzf_t *b, record_t *rec)
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
	rec->c.tid = (int32_t)le32(core + 0);
	rec->c.pos = (int32_t)le32(core + 4);
	rec->l_read_name = core[8];
	rec->mapq = core[9];
	rec->n_cigar = le16(core + 12);
	rec->c.flag = le16(core + 14);
	rec->l_seq = le32(core + 16);
	rec->c.mtid = (int32_t)le32(core + 20);
	rec->c.mpos = (int32_t)le32(core + 24);
	rec->c.tlen = (int32_t)le32(core + 28);
	uint32_t rest = block_size - 32;
	/* htslib bam_read1: a record whose variable part cannot hold its own fields is invalid */
	if (rec->l_read_name < 1 || rec->l_seq > 0x7fffffffu ||
	    (uint64_t)rec->n_cigar * 4 + rec->l_read_name + (((uint64_t)rec->l_seq + 1) >> 1) + rec->l_seq > rest)
		oc_error("Invalid BAM record.");
	oc_clear(&rec->body);
	oc_reserve(&rec->body, (size_t)rest + 1);
	r = bgzf_read(b, rec->body.p, rest);
	if (r < 0) oc_error("Invalid BAM record.");
	if (r != (long)rest) oc_error("BAM file ended prematurely.");
	rec->body.n = rest;
	return 1;
}

/* HashMap<Box<str>, Box<str>> with insert / remove; iteration in insertion order (the
 * reference's order is arbitrary; see seqkit_oracle.h, orc_census)                       */
typedef struct { oc_str key, val; uint64_t seq; int live; } pend_t;
typedef struct { pend_t *ent; size_t n, cap; int64_t *slot; size_t nslot; size_t nlive; uint64_t next_seq; } pendmap;

static void pendmap_rehash(pendmap *m, size_t nslot)
{
	/* drop dead entries, then index the live ones */
	size_t w = 0;
	for (size_t e = 0; e < m->n; e++) {
		if (m->ent[e].live) { if (w != e) { pend_t t = m->ent[w]; m->ent[w] = m->ent[e]; m->ent[e] = t; } w++; }
	}
	for (size_t e = w; e < m->n; e++) { free(m->ent[e].key.p); free(m->ent[e].val.p); memset(&m->ent[e], 0, sizeof(pend_t)); }
	m->n = w;
	free(m->slot);
	m->slot = (int64_t *)malloc(nslot * sizeof *m->slot);
	m->nslot = nslot;
	for (size_t i = 0; i < nslot; i++) m->slot[i] = -1;
	for (size_t e = 0; e < m->n; e++) {
		size_t i = oc_hash(m->ent[e].key.p, m->ent[e].key.n) & (nslot - 1);
		while (m->slot[i] >= 0) i = (i + 1) & (nslot - 1);
		m->slot[i] = (int64_t)e;
	}
}
static pend_t *pendmap_find(pendmap *m, const uint8_t *key, size_t n)
{
	if (m->nslot == 0) return NULL;
	size_t i = oc_hash(key, n) & (m->nslot - 1);
	while (m->slot[i] >= 0) {
		pend_t *e = &m->ent[m->slot[i]];
		if (e->live && e->key.n == n && memcmp(e->key.p, key, n) == 0) return e;
		i = (i + 1) & (m->nslot - 1);
	}
	return NULL;
}
static void pendmap_insert(pendmap *m, const uint8_t *key, size_t kn, const oc_str *val)
{
	pend_t *e = pendmap_find(m, key, kn);
	if (e) { oc_assign(&e->val, val->p, val->n); return; }            /* HashMap::insert replaces the value */
	if (m->nslot == 0 || (m->n + 1) * 2 > m->nslot) pendmap_rehash(m, m->nslot ? (m->nlive * 4 > m->nslot ? m->nslot * 2 : m->nslot) : 1024);
	if (m->n == m->cap) {
		m->cap = m->cap ? m->cap * 2 : 256;
		m->ent = (pend_t *)realloc(m->ent, m->cap * sizeof *m->ent);
		memset(m->ent + m->n, 0, (m->cap - m->n) * sizeof *m->ent);
	}
	e = &m->ent[m->n];
	oc_assign(&e->key, key, kn);
	oc_assign(&e->val, val->p, val->n);
	e->seq = m->next_seq++;
	e->live = 1;
	size_t i = oc_hash(key, kn) & (m->nslot - 1);
	while (m->slot[i] >= 0) i = (i + 1) & (m->nslot - 1);
	m->slot[i] = (int64_t)m->n++;
	m->nlive++;
}
static void pendmap_remove(pendmap *m, pend_t *e) { e->live = 0; m->nlive--; }

typedef enum { RAW, FASTA, FASTQ } outfmt;

static int is_char_boundary(const oc_str *s, size_t i) { return i == s->n || (i < s->n && (s->p[i] & 0xC0) != 0x80); }

/* src/sam_to_fastq.rs:138-149; out == NULL is io::sink() */
static void write_read(FILE *out, outfmt format, const uint8_t *qname, size_t qn, const oc_str *seq)
{
	if (format == FASTQ) {
		size_t seq_len = (seq->n - 1) / 2;                                       /* :141 */
		if (!is_char_boundary(seq, seq_len) || !is_char_boundary(seq, seq_len + 1)) oc_panic("byte index is not a char boundary");
		if (!out) return;
		fputc('@', out); fwrite(qname, 1, qn, out); fputc('\n', out);
		fwrite(seq->p, 1, seq_len, out); fputs("\n+\n", out);
		fwrite(seq->p + seq_len + 1, 1, seq->n - seq_len - 1, out); fputc('\n', out);
	} else if (format == FASTA) {
		if (!out) return;
		fputc('>', out); fwrite(qname, 1, qn, out); fputc('\n', out);
		fwrite(seq->p, 1, seq->n, out); fputc('\n', out);
	} else {
		if (!out) return;
		fwrite(seq->p, 1, seq->n, out); fputc('\n', out);
	}
}

static int to_reads(int argc, char **argv)
{
	/* docopt: `sam to (raw|fasta|fastq) <bam_file> <out_prefix>` | `sam to interleaved (raw|fasta|fastq) <bam_file>` */
	int interleaved = argc >= 4 && !strcmp(argv[2], "interleaved");
	const char *fmtw = argv[interleaved ? 3 : 2];
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, interleaved ? 4 : 3, NULL, 0, pos, &npos, 2) || npos != (interleaved ? 1 : 2))
		oc_error("Invalid arguments.\n%s", USAGE_TO);
	outfmt format = !strcmp(fmtw, "raw") ? RAW : !strcmp(fmtw, "fasta") ? FASTA : FASTQ;      /* :68-71 */
	FILE *out_1, *out_2, *out_single;
	if (interleaved) { out_1 = stdout; out_2 = stdout; out_single = NULL; }                   /* :74-78 */
	else {                                                                                     /* :79-86 */
		const char *ext = format == RAW ? "seq" : format == FASTA ? "fa" : "fq";
		char path[4096];
		snprintf(path, sizeof path, "%s_1.%s.gz", pos[1], ext); out_1 = oc_gzip_writer(path, 0);
		snprintf(path, sizeof path, "%s_2.%s.gz", pos[1], ext); out_2 = oc_gzip_writer(path, 0);
		snprintf(path, sizeof path, "%s.%s.gz", pos[1], ext); out_single = oc_gzip_writer(path, 0);
	}
	bgzf_t *b = (bgzf_t *)malloc(sizeof(bgzf_t));                                             /* :96 */
	bgzf_open(b, pos[0]);
	bam_read_header(b, pos[0]);
	pendmap reads_1 = {0}, reads_2 = {0};
	record_t rec;
	memset(&rec, 0, sizeof rec);
	oc_str read_seq = {0}, bases = {0};
	while (bam_next_full(b, &rec)) {                                                           /* :101 */
		if (rec.c.flag & 0x100 || rec.c.flag & 0x800) continue;                               /* :102 */
		const uint8_t *qname = rec.body.p;
		size_t qn = rec.l_read_name - 1;                                                       /* rust-htslib qname(): without the final NUL */
		if (!orc_utf8_valid(qname, qn)) oc_panic("called `Result::unwrap()` on an `Err` value: Utf8Error");   /* :104 */
		const uint8_t *seq4 = rec.body.p + rec.l_read_name + 4 * (size_t)rec.n_cigar;
		const uint8_t *qual = seq4 + (rec.l_seq + 1) / 2;
		oc_clear(&bases);
		oc_reserve(&bases, rec.l_seq + 1);
		orc_bam_sequence(seq4, qual, rec.l_seq, (rec.c.flag & 0x10) != 0, 10, bases.p);        /* :105 */
		oc_assign(&read_seq, bases.p, rec.l_seq);
		if (format == FASTQ) {                                                                 /* :107-112 */
			oc_append(&read_seq, "|", 1);
			for (uint32_t k = 0; k < rec.l_seq; k++) {
				uint8_t ch = (uint8_t)(33 + qual[k]);                                          /* u8 arithmetic wraps (release build) */
				if (ch < 0x80) oc_append(&read_seq, &ch, 1);
				else { uint8_t u[2] = {(uint8_t)(0xC0 | (ch >> 6)), (uint8_t)(0x80 | (ch & 0x3F))}; oc_append(&read_seq, u, 2); }   /* char::from(u8) pushed as UTF-8 */
			}
		}
		if (!(rec.c.flag & 0x1)) {                                                             /* :114-115 */
			write_read(out_single, format, qname, qn, &read_seq);
		} else if (rec.c.flag & 0x40) {                                                        /* :116-122 */
			pend_t *mate = pendmap_find(&reads_2, qname, qn);
			if (mate) {
				write_read(out_1, format, qname, qn, &read_seq);
				write_read(out_2, format, qname, qn, &mate->val);
				pendmap_remove(&reads_2, mate);
			} else pendmap_insert(&reads_1, qname, qn, &read_seq);
		} else if (rec.c.flag & 0x80) {                                                        /* :123-130 */
			pend_t *mate = pendmap_find(&reads_1, qname, qn);
			if (mate) {
				write_read(out_1, format, qname, qn, &mate->val);
				write_read(out_2, format, qname, qn, &read_seq);
				pendmap_remove(&reads_1, mate);
			} else pendmap_insert(&reads_2, qname, qn, &read_seq);
		}
	}
	/* :133-137 orphans: reads_1 then reads_2 */
	pendmap *maps[2] = {&reads_1, &reads_2};
	for (int k = 0; k < 2; k++)
		for (size_t e = 0; e < maps[k]->n; e++)                                                /* entries are stored in insertion order */
			if (maps[k]->ent[e].live) write_read(out_single, format, maps[k]->ent[e].key.p, maps[k]->ent[e].key.n, &maps[k]->ent[e].val);
	return 0;
}

/* ---- sam count (src/sam_count.rs:20-130) ----------------------------------------------------- */
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

/* record with mapq and cigar end_pos (rust-htslib CigarStringView::end_pos: pos + reference-consuming ops) */
static int bam_next_count(bgzf_t *b, core_t *c, uint8_t *mapq, int32_t *end_pos)
{
	static record_t rec;
	if (!bam_next_full(b, &rec)) return 0;
	*c = rec.c;
	*mapq = rec.mapq;
	int64_t e = rec.c.pos;
	for (uint32_t k = 0; k < rec.n_cigar; k++) {
		uint32_t op = le32(rec.body.p + rec.l_read_name + 4 * k);
		uint32_t code = op & 15, len = op >> 4;
		if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) e += len;
	}
	*end_pos = (int32_t)e;
	return 1;
}

static int count(int argc, char **argv)
{
	oc_opt opts[4] = {{"--min-mapq", 1, NULL}, {"--max-frag-len", 1, NULL}, {"--single-end", 0, NULL}, {"--center", 0, NULL}};
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 2, opts, 4, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_COUNT);
	uint64_t v;
	orc_count_params p;
	if (!oc_parse_uint(opts[0].value ? opts[0].value : "0", 255, &v)) oc_error("--min-mapq must be an integer between 0 - 255.");   /* :23-24 */
	p.min_mapq = (uint8_t)v;
	if (!oc_parse_uint(opts[1].value ? opts[1].value : "5000", 0xffffffffull, &v)) oc_error("--max-frag-len must be an integer.");   /* :25 */
	p.max_frag_len = (uint32_t)v;
	p.single_end = opts[2].value != NULL;                                      /* :26 */
	p.count_centers = opts[3].value != NULL;                                   /* :27 */

	fputs("Reading target regions from BED file...\n", stderr);                /* :30 */
	/* read_regions, src/common.rs:198-219 */
	oc_str *rchr_name = NULL; uint32_t *rstart = NULL, *rend = NULL; size_t nreg = 0, capreg = 0;
	{
		oc_reader bed = oc_reader_open(pos[1]);
		oc_str line = {0};
		while (oc_read_line(&bed, &line)) {
			if (oc_starts_with(&line, '#')) continue;
			size_t off = orc_trim_start_off(line.p, line.n), end = orc_trim_end_len(line.p, line.n);
			if (end < off) end = off;
			/* split('\t') */
			size_t col_st[3] = {0, 0, 0}, col_en[3] = {0, 0, 0}; int ncol = 0;
			size_t a = off;
			for (;;) {
				size_t b2 = a;
				while (b2 < end && line.p[b2] != '\t') b2++;
				if (ncol < 3) { col_st[ncol] = a; col_en[ncol] = b2; }
				ncol++;
				if (b2 >= end) break;
				a = b2 + 1;
			}
			if (ncol < 3) oc_error("Invalid region in BED file:\n%s", (const char *)line.p);
			if (nreg == capreg) {
				capreg = capreg ? capreg * 2 : 256;
				rchr_name = (oc_str *)realloc(rchr_name, capreg * sizeof(oc_str));
				rstart = (uint32_t *)realloc(rstart, capreg * 4);
				rend = (uint32_t *)realloc(rend, capreg * 4);
			}
			memset(&rchr_name[nreg], 0, sizeof(oc_str));
			oc_assign(&rchr_name[nreg], line.p + col_st[0], col_en[0] - col_st[0]);
			char num[64];
			uint64_t s, e;
			size_t l1 = col_en[1] - col_st[1], l2 = col_en[2] - col_st[2];
			if (l1 >= sizeof num || l2 >= sizeof num) oc_panic("called `Result::unwrap()` on an `Err` value: ParseIntError");
			memcpy(num, line.p + col_st[1], l1); num[l1] = 0;
			if (!oc_parse_uint(num, 0xffffffffull, &s)) oc_panic("called `Result::unwrap()` on an `Err` value: ParseIntError");
			memcpy(num, line.p + col_st[2], l2); num[l2] = 0;
			if (!oc_parse_uint(num, 0xffffffffull, &e)) oc_panic("called `Result::unwrap()` on an `Err` value: ParseIntError");
			rstart[nreg] = (uint32_t)s; rend[nreg] = (uint32_t)e;
			nreg++;
		}
	}
	uint32_t *region_frags = (uint32_t *)calloc(nreg ? nreg : 1, 4);           /* :32 */
	fprintf(stderr, "Counting %s...\n", p.single_end ? "reads" : "DNA fragments");   /* :34-35 */
	bgzf_t *b = (bgzf_t *)malloc(sizeof(bgzf_t));                              /* :36 */
	bgzf_open(b, pos[0]);
	bam_read_header(b, pos[0]);
	for (uint32_t i = 0; i < g_n_ref; i++)                                     /* :37-38 from_utf8(name).unwrap() */
		if (!orc_utf8_valid((const uint8_t *)g_ref_names[i], strlen(g_ref_names[i]))) oc_panic("called `Result::unwrap()` on an `Err` value: Utf8Error");
	/* region -> BAM reference by name; a region list per reference is rebuilt whenever the chromosome changes (:52-67),
	 * which orc_count_record does from rchr.  (With duplicate reference names only the first one gets the regions.) */
	int32_t *rchr = (int32_t *)malloc((nreg ? nreg : 1) * 4);
	for (size_t r = 0; r < nreg; r++) {
		rchr[r] = -1;
		for (uint32_t i = 0; i < g_n_ref; i++)
			if (strlen(g_ref_names[i]) == rchr_name[r].n && memcmp(g_ref_names[i], rchr_name[r].p, rchr_name[r].n) == 0) { rchr[r] = (int32_t)i; break; }
	}
	orc_count_state st;
	orc_count_state_init(&st);
	core_t c; uint8_t mapq; int32_t end_pos;
	while (bam_next_count(b, &c, &mapq, &end_pos)) {                           /* :45 */
		int rc = orc_count_record(&st, c.flag, mapq, c.tid, c.mtid, c.pos, c.mpos, c.tlen, end_pos, &p, (int32_t)g_n_ref,
		                          rchr, rstart, rend, (int64_t)nreg, region_frags);
		if (rc == ORC_COUNT_UNSORTED) oc_error("Input BAM file is not coordinate sorted.");
		if (rc == ORC_COUNT_BAD_TID) oc_panic("index out of bounds: chr_names[tid]");
	}
	for (size_t r = 0; r < nreg; r++) printf("%u\n", region_frags[r]);         /* :128-130 */
	return 0;
}

int main(int argc, char **argv)
{
	int rc;
	if (argc >= 2 && !strcmp(argv[1], "fragments")) rc = fragments(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "statistics")) rc = statistics(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "count")) rc = count(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "fragment") && !strcmp(argv[2], "lengths")) rc = fragment_lengths(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "to") && (!strcmp(argv[2], "raw") || !strcmp(argv[2], "fasta") || !strcmp(argv[2], "fastq"))) rc = to_reads(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "to") && !strcmp(argv[2], "interleaved") &&
	         (!strcmp(argv[3], "raw") || !strcmp(argv[3], "fasta") || !strcmp(argv[3], "fastq"))) rc = to_reads(argc, argv);
	else { fprintf(stderr, "%s\n", USAGE_TOP); rc = 0; }
	fflush(stdout);
	oc_wait_children();
	return rc;
}
