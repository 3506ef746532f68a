/*
 * fasta_oracle — ORACLE restatement of the reference's `fasta` subcommands on the hot
 * path, line-at-a-time, single-threaded, same loops as the cited source.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see seqkit_oracle.h).
 *
 *   fasta trim by quality <fastq_file> <min_baseq>      src/fasta_trim_by_quality.rs:10-50
 *   fasta mask by quality <fastq_file> <min_baseq>      src/fasta_mask_by_quality.rs:11-47
 *   fasta add barcode <fastq_file> <barcode_file>       src/fasta_add_barcode.rs:11-45
 *   fasta demultiplex [options] <sheet> <fq1> [<fq2>]   src/fasta_demultiplex.rs:30-265
 *   fasta statistics <fastq_file>                        src/fasta_statistics.rs:12-51
 *   fasta gc content <genome.fa> <regions.bed>           src/fasta_gc_content.rs:17-50
 *   and the line filters: check, to raw, add / remove base qualities, simplify read ids, interleave, deinterleave,
 *   split into anchors, trim, extract dual umi, convert basespace (each function cites its source file)
 * dispatch: src/fasta_main.rs:42-82.
 */
#include "cli_common.h"

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
static const char *USAGE_STATS = "\nUsage:\n  fasta statistics <fastq_file>\n";
static const char *USAGE_MASK = "\nUsage:\n  fasta mask by quality <fastq_file> <min_baseq>\n";
static const char *USAGE_ADDBC = "\nUsage:\n  fasta add barcode <fastq_file> <barcode_file>\n";
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

/* ---------------------------------------------------------------------------------- */
/* src/fasta_trim_by_quality.rs:10-50 */
static int trim_by_quality(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 4, NULL, 0, pos, &npos, 2) || npos != 2)
		oc_error("Invalid arguments.\n%s", USAGE_TRIM);
	oc_reader fq = oc_reader_open(pos[0]);                                     /* :12 */
	uint64_t mb;
	if (!oc_parse_uint(pos[1], 255, &mb)) oc_panic("min_baseq.parse::<u8>().unwrap()"); /* :13 */
	uint8_t min_baseq = (uint8_t)mb;

	oc_str line = {0}, seq = {0}, qual = {0};
	while (oc_read_line(&fq, &line)) {                                         /* :19 */
		if (!oc_starts_with(&line, '@')) oc_error("Invalid FASTQ format encountered."); /* :20-22 */
		fwrite(line.p, 1, line.n, stdout);                                     /* :23 */
		oc_read_line(&fq, &seq);                                               /* :24 */
		oc_read_line(&fq, &line);                                              /* :25 */
		oc_read_line(&fq, &qual);                                              /* :26 */

		uint32_t n = (uint32_t)orc_trim_end_len(qual.p, qual.n);               /* :31 */
		uint32_t lowest_k = orc_trim_lowest_k(qual.p, n, min_baseq);           /* :28-42 */

		if (lowest_k == 0) {                                                   /* :44-45 */
			fputs("N\n+\n!\n", stdout);
		} else {                                                               /* :47 */
			/* &seq[..lowest_k] / &qual[..lowest_k]: byte slices; out of range or a cut
			 * inside a multi-byte char panics. */
			if (lowest_k > seq.n) oc_panic("byte index out of range of `seq`");
			if (lowest_k < seq.n && (seq.p[lowest_k] & 0xC0) == 0x80) oc_panic("seq slice not on a char boundary");
			if (lowest_k < qual.n && (qual.p[lowest_k] & 0xC0) == 0x80) oc_panic("qual slice not on a char boundary");
			fwrite(seq.p, 1, lowest_k, stdout);
			fputs("\n+\n", stdout);
			fwrite(qual.p, 1, lowest_k, stdout);
			fputc('\n', stdout);
		}
	}
	return 0;
}

/* ---------------------------------------------------------------------------------- */
static size_t u8_char_len(uint8_t b) { return b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : 4; }
static uint32_t u8_decode(const uint8_t *s, size_t l)
{
	if (l == 1) return s[0];
	if (l == 2) return ((uint32_t)(s[0] & 0x1F) << 6) | (s[1] & 0x3F);
	if (l == 3) return ((uint32_t)(s[0] & 0x0F) << 12) | ((uint32_t)(s[1] & 0x3F) << 6) | (s[2] & 0x3F);
	return ((uint32_t)(s[0] & 0x07) << 18) | ((uint32_t)(s[1] & 0x3F) << 12) | ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
}

/* src/fasta_mask_by_quality.rs:11-47 */
static int mask_by_quality(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 4, NULL, 0, pos, &npos, 2) || npos != 2)
		oc_error("Invalid arguments.\n%s", USAGE_MASK);
	oc_reader fq = oc_reader_open(pos[0]);                                     /* :13 */
	uint64_t mb;
	if (!oc_parse_uint(pos[1], 255, &mb)) oc_panic("min_baseq.parse::<u8>().unwrap()"); /* :14 */
	uint8_t min_baseq = (uint8_t)mb;

	oc_str seq = {0}, bq = {0}, line = {0}, output = {0};
	while (oc_read_line(&fq, &line)) {                                         /* :20 */
		if (!oc_starts_with(&line, '@')) oc_error("Invalid FASTQ format encountered."); /* :21-23 */
		oc_assign(&output, line.p, line.n);                                    /* :25-26 */
		oc_read_line(&fq, &seq);                                               /* :28 */
		oc_read_line(&fq, &line);                                              /* :29 */
		oc_read_line(&fq, &bq);                                                /* :30 */
		if (seq.n && seq.p[seq.n - 1] == '\n') seq.n--;                        /* :32 */
		if (bq.n && bq.p[bq.n - 1] == '\n') bq.n--;                            /* :33 */
		if (seq.n != bq.n)                                                     /* :35-37 */
			oc_error("Read sequence and base qualities are of different length.");
		/* :40-43 — seq.chars().zip(qual.chars()); `qual as u8` keeps the low 8 bits of
		 * the code point; for pure-ASCII lines this is the byte loop of orc_mask_bytes. */
		size_t i = 0, j = 0;
		while (i < seq.n && j < bq.n) {
			size_t li = u8_char_len(seq.p[i]), lj = u8_char_len(bq.p[j]);
			uint8_t q = (uint8_t)u8_decode(bq.p + j, lj);
			if ((uint8_t)(q - (uint8_t)33) < min_baseq) oc_append(&output, "N", 1);
			else oc_append(&output, seq.p + i, li);
			i += li; j += lj;
		}
		oc_append(&output, "\n+\n", 3);                                        /* :44 */
		oc_append(&output, bq.p, bq.n);
		oc_append(&output, "\n", 1);
		fwrite(output.p, 1, output.n, stdout);                                 /* :45 */
	}
	return 0;
}

/* ---------------------------------------------------------------------------------- */
/* src/fasta_add_barcode.rs:11-45 */
static int add_barcode(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 3, NULL, 0, pos, &npos, 2) || npos != 2)
		oc_error("Invalid arguments.\n%s", USAGE_ADDBC);
	oc_reader fq = oc_reader_open(pos[0]);
	oc_reader bf = oc_reader_open(pos[1]);
	oc_str header = {0}, barcode = {0}, line = {0};
	for (;;) {
		oc_read_line(&bf, &header);                                            /* :20 */
		if (oc_starts_with(&header, '@')) {                                    /* :21-24 */
			oc_read_line(&bf, &barcode);
			oc_read_line(&bf, &line);
			oc_read_line(&bf, &line);
		} else if (oc_starts_with(&header, '>')) {                             /* :25-27 */
			oc_read_line(&bf, &barcode);
		}                              /* exhausted barcode file: `barcode` keeps its old value */
		if (!oc_read_line(&fq, &header)) break;                                /* :29-31 */
		size_t hl = orc_trim_end_len(header.p, header.n);
		size_t bl = orc_trim_end_len(barcode.p, barcode.n);
		fwrite(header.p, 1, hl, stdout);                                       /* :33 */
		fputs(" BC:", stdout);
		fwrite(barcode.p, 1, bl, stdout);
		fputc('\n', stdout);
		if (oc_starts_with(&header, '@')) {                                    /* :35-38 */
			for (int k = 0; k < 3; k++) { oc_read_line(&fq, &line); fwrite(line.p, 1, line.n, stdout); }
		} else if (oc_starts_with(&header, '>')) {                             /* :39-40 */
			oc_read_line(&fq, &line); fwrite(line.p, 1, line.n, stdout);
		} else {                                                               /* :41-43 */
			oc_error("Invalid FASTQ line:\n%s", (const char *)header.p);
		}
	}
	return 0;
}

/* ---------------------------------------------------------------------------------- */
typedef struct {
	oc_str name, barcode;
	FILE *out[2];
	uint64_t total_reads;
} sample_t;


/* `entries.sort_by_key(|x| x.1); entries.reverse(); for ... in &entries[0..100]`
 * (src/fasta_statistics.rs:45-50, src/fasta_demultiplex.rs:255-260).  entries come in as
 * samples (sheet order) then map entries (HashMap order: arbitrary in the reference,
 * first-seen order here).  Stable ascending sort by count (merge sort, like Rust's), then
 * reverse().                                                                            */
typedef struct { const uint8_t *label; uint64_t count; } oc_entry;

static void entry_merge_sort(oc_entry *a, oc_entry *tmp, size_t n)
{
	if (n < 2) return;
	size_t h = n / 2;
	entry_merge_sort(a, tmp, h);
	entry_merge_sort(a + h, tmp, n - h);
	size_t i = 0, j = h, k = 0;
	while (i < h && j < n) tmp[k++] = (a[j].count < a[i].count) ? a[j++] : a[i++];
	while (i < h) tmp[k++] = a[i++];
	while (j < n) tmp[k++] = a[j++];
	memcpy(a, tmp, n * sizeof *a);
}

static void print_most_frequent(oc_entry *ents, size_t ne)
{
	oc_entry *tmp = (oc_entry *)malloc(sizeof(oc_entry) * (ne ? ne : 1));
	entry_merge_sort(ents, tmp, ne);
	free(tmp);
	/* &entries[0..100] panics when there are fewer than 100 entries */
	if (ne < 100) oc_panic("range end index 100 out of range for slice");
	for (size_t a = 0; a < 100; a++) {
		oc_entry *x = &ents[ne - 1 - a];
		printf("- %s: %llu\n", (const char *)x->label, (unsigned long long)x->count);
	}
}

/* src/fasta_statistics.rs:12-51 */
static int statistics(int argc, char **argv)
{
	const char *pos[1];
	int npos = 0;
	if (!oc_parse(argc, argv, 2, NULL, 0, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_STATS);
	oc_reader fastq = oc_reader_open(pos[0]);                                   /* :14 */
	uint64_t total_records = 0;
	oc_countmap sample_barcodes = {0};
	oc_str line = {0}, skip = {0};
	while (oc_read_line(&fastq, &line)) {                                      /* :22 */
		size_t st, en;
		if (orc_find_bc_field_stats(line.p, line.n, &st, &en))                 /* :24-27 */
			oc_countmap_add(&sample_barcodes, line.p + st + 4, en - st - 4);
		if (oc_starts_with(&line, '@')) { for (int k = 0; k < 3; k++) oc_read_line(&fastq, &skip); }    /* :30-31 */
		else if (oc_starts_with(&line, '>')) oc_read_line(&fastq, &skip);      /* :32-33 */
		else oc_error("Invalid FASTQ header:\n%s", (const char *)line.p);      /* :34-36 */
		total_records += 1;                                                    /* :38 */
	}
	printf("Total sequence records: %llu\n", (unsigned long long)total_records);        /* :41 */
	printf("Most frequent sample barcodes:\n");                                /* :43 */
	oc_entry *ents = (oc_entry *)malloc(sizeof(oc_entry) * (sample_barcodes.n ? sample_barcodes.n : 1));
	for (size_t e = 0; e < sample_barcodes.n; e++) { ents[e].label = sample_barcodes.ent[e].key.p; ents[e].count = sample_barcodes.ent[e].count; }
	print_most_frequent(ents, sample_barcodes.n);
	return 0;
}

/* src/fasta_demultiplex.rs:30-265 */
static int demultiplex(int argc, char **argv)
{
	oc_opt opts[4] = {
		{"--parallel", 0, NULL}, {"--index1", 1, NULL}, {"--index2", 1, NULL}, {"--dry-run", 1, NULL}};
	const char *pos[3]; int npos;
	if (!oc_parse(argc, argv, 2, opts, 4, pos, &npos, 3) || npos < 2)
		oc_error("Invalid arguments.\n%s", USAGE_DEMUX);
	int parallel = opts[0].value != NULL;                                      /* :32 */
	uint64_t dry_run = 0;                                                      /* :33-36 */
	const char *dr = opts[3].value ? opts[3].value : "";
	if (!oc_parse_uint(dr, UINT64_MAX, &dry_run)) dry_run = 0;
	if (dry_run == 0 && dr[0] != 0) oc_error("In --dry-run=N, N must be 64-bit positive integer.");

	oc_reader fastq[2]; int nfastq = 0;                                        /* :41-46 */
	fastq[nfastq++] = oc_reader_open(pos[1]);
	if (npos == 3 && pos[2][0] != 0) fastq[nfastq++] = oc_reader_open(pos[2]);
	int paired_end = nfastq == 2;

	oc_reader index_fastq[2]; int nindex = 0;                                  /* :49-55 */
	if (opts[1].value && opts[1].value[0]) index_fastq[nindex++] = oc_reader_open(opts[1].value);
	if (opts[2].value && opts[2].value[0]) index_fastq[nindex++] = oc_reader_open(opts[2].value);

	fputs("Reading sample sheet...\n", stderr);                                /* :58 */
	oc_reader sheet = oc_reader_open(pos[0]);
	sample_t *samples = NULL; int S = 0, capS = 0;
	oc_str line = {0};
	size_t barcode_len = 0;
	while (oc_read_line(&sheet, &line)) {                                      /* :63 */
		if (oc_starts_with(&line, '#')) continue;                              /* :64 */
		size_t off = orc_trim_start_off(line.p, line.n);                       /* :65 line.trim() */
		size_t tl = orc_trim_end_len(line.p + off, line.n - off);
		const uint8_t *t = line.p + off;
		/* split('\t'): cols[0], cols[1] */
		const uint8_t *tab1 = (const uint8_t *)memchr(t, '\t', tl);
		if (!tab1) continue;                                                   /* :66 cols.len() < 2 */
		size_t c0 = (size_t)(tab1 - t);
		const uint8_t *c1p = tab1 + 1;
		size_t rest = tl - c0 - 1;
		const uint8_t *tab2 = (const uint8_t *)memchr(c1p, '\t', rest);
		size_t c1 = tab2 ? (size_t)(tab2 - c1p) : rest;
		if (S == capS) { capS = capS ? capS * 2 : 16; samples = (sample_t *)realloc(samples, sizeof(sample_t) * capS); }
		sample_t *sm = &samples[S];
		memset(sm, 0, sizeof(*sm));
		oc_assign(&sm->name, t, c0);
		if (c1 == 0) oc_error("Sample %s has no barcode.", (const char *)sm->name.p);     /* :68 */
		if (barcode_len == 0) barcode_len = c1;                                /* :69-70 */
		else if (c1 != barcode_len) oc_error("Barcodes in sample sheet must all be of same length."); /* :71-73 */
		oc_assign(&sm->barcode, c1p, c1);
		if (dry_run > 0) {                                                     /* :77-78 */
		} else if (paired_end) {                                               /* :79-83 */
			char path[4096];
			snprintf(path, sizeof path, "%s_1.fq.gz", (const char *)sm->name.p);
			sm->out[0] = oc_gzip_writer(path, parallel);
			snprintf(path, sizeof path, "%s_2.fq.gz", (const char *)sm->name.p);
			sm->out[1] = oc_gzip_writer(path, parallel);
		} else {                                                               /* :84-87 */
			char path[4096];
			snprintf(path, sizeof path, "%s.fq.gz", (const char *)sm->name.p);
			sm->out[0] = oc_gzip_writer(path, parallel);
		}
		S++;
	}

	for (int s = 0; s < S; s++)                                                /* :98-104 */
		for (int k = s + 1; k < S; k++)
			if (samples[s].name.n == samples[k].name.n &&
			    memcmp(samples[s].name.p, samples[k].name.p, samples[s].name.n) == 0)
				oc_error("Sample %s is listed multiple times in sample sheet.", (const char *)samples[s].name.p);

	fprintf(stderr, "Starting demultiplexing in %s end mode...\n", paired_end ? "paired" : "single"); /* :106-107 */
	uint64_t total_reads = 0, identified_reads = 0;
	oc_countmap extra_barcodes = {0};                                          /* :110 */

	oc_str header = {0}, barcode = {0}, umi = {0};
	while (oc_read_line(&fastq[0], &header)) {                                 /* :117 */
		if (!oc_starts_with(&header, '@'))                                     /* :118-120 */
			oc_error("Invalid FASTQ header line:\n%s", (const char *)header.p);
		oc_clear(&barcode);                                                    /* :123 */
		if (nindex > 0) {                                                      /* :126-136 */
			for (int f = 0; f < nindex; f++) {
				if (barcode.n) oc_append(&barcode, "+", 1);
				oc_read_line(&index_fastq[f], &line);
				if (!oc_starts_with(&line, '@')) oc_panic("assertion failed: line.starts_with('@')");
				oc_read_line(&index_fastq[f], &line);
				oc_append(&barcode, line.p, orc_trim_end_len(line.p, line.n));
				oc_read_line(&index_fastq[f], &line);
				if (!oc_starts_with(&line, '+')) oc_panic("assertion failed: line.starts_with('+')");
				oc_read_line(&index_fastq[f], &line);
			}
		} else {                                                               /* :137-146 */
			size_t st, en;
			if (!orc_find_bc_field(header.p, header.n, &st, &en)) oc_error("No BC:xxxx field found.");
			oc_append(&barcode, header.p + st + 4, en - (st + 4));
			oc_drain(&header, st, en);
		}
		if (barcode.n != barcode_len)                                          /* :148-150 */
			oc_error("Sequenced barcode %s is of different length (%zu nt) than barcodes in the sample sheet (%zu nt).",
			         barcode.n ? (const char *)barcode.p : "", barcode.n, barcode_len);

		/* :154-166 — restated inline (the table is not contiguous here) */
		size_t best_sample = 0, equally_fine_sample = 0;
		uint64_t lowest_diff = UINT64_MAX;
		for (int s = 0; s < S; s++) {
			uint64_t diff = orc_barcode_diff(barcode.p, samples[s].barcode.p, barcode.n);
			if (diff < lowest_diff) { lowest_diff = diff; best_sample = (size_t)s; equally_fine_sample = (size_t)s; }
			else if (diff == lowest_diff) equally_fine_sample = (size_t)s;
		}

		const uint64_t MAX_BARCODE_DIFFERENCE = 1;                             /* :168 */
		total_reads += 1;                                                      /* :169 */
		int write_read_out = 0;
		if (lowest_diff <= MAX_BARCODE_DIFFERENCE) {                           /* :172 */
			if (best_sample == equally_fine_sample) {                          /* :173-179 */
				identified_reads += 1;
				samples[best_sample].total_reads += 1;
				write_read_out = !(dry_run > 0);
			} else {                                                           /* :181-189 */
				fprintf(stderr, "WARNING: Sequenced barcode %s was an equally good match (%llu mismatches) for samples %s (%s) and %s (%s), and was therefore not assigned to any sample.\n",
				        (const char *)barcode.p, (unsigned long long)lowest_diff,
				        (const char *)samples[best_sample].name.p, (const char *)samples[best_sample].barcode.p,
				        (const char *)samples[equally_fine_sample].name.p, (const char *)samples[equally_fine_sample].barcode.p);
			}
		} else if (dry_run > 0) {                                              /* :190-194 */
			oc_countmap_add(&extra_barcodes, barcode.p, barcode.n);
		}

		if (write_read_out) {                                                  /* :196 */
			sample_t *sm = &samples[best_sample];
			oc_clear(&umi);                                                    /* :200-203 (chars().zip(chars())) */
			{
				size_t i = 0, j = 0;
				while (i < sm->barcode.n && j < barcode.n) {
					size_t li = u8_char_len(sm->barcode.p[i]), lj = u8_char_len(barcode.p[j]);
					if (li == 1 && sm->barcode.p[i] == 'U') oc_append(&umi, barcode.p + j, lj);
					i += li; j += lj;
				}
			}
			fwrite(header.p, 1, orc_trim_end_len(header.p, header.n), sm->out[0]);   /* :206 */
			if (umi.n) { fputs(" UMI:", sm->out[0]); fwrite(umi.p, 1, umi.n, sm->out[0]); } /* :207 */
			fputc('\n', sm->out[0]);                                           /* :208 */
			for (int k = 0; k < 3; k++) {                                      /* :209-212 */
				oc_read_line(&fastq[0], &line);
				fwrite(line.p, 1, line.n, sm->out[0]);
			}
			if (paired_end) {                                                  /* :215 */
				oc_read_line(&fastq[1], &line);                                /* :216 */
				if (nindex == 0) {                                             /* :219-227 */
					size_t st, en;
					if (orc_find_bc_field(line.p, line.n, &st, &en) && en > 0) oc_drain(&line, st, en);
				}
				fwrite(line.p, 1, orc_trim_end_len(line.p, line.n), sm->out[1]);      /* :229 */
				if (umi.n) { fputs(" UMI:", sm->out[1]); fwrite(umi.p, 1, umi.n, sm->out[1]); } /* :230-232 */
				fputc('\n', sm->out[1]);                                       /* :233 */
				for (int k = 0; k < 3; k++) {                                  /* :234-237 */
					oc_read_line(&fastq[1], &line);
					fwrite(line.p, 1, line.n, sm->out[1]);
				}
			}
		} else {                                                               /* :239-246 */
			for (int k = 0; k < 3; k++) oc_read_line(&fastq[0], &line);
			if (paired_end) for (int k = 0; k < 4; k++) oc_read_line(&fastq[1], &line);
		}
		if (dry_run > 0 && total_reads >= dry_run) break;                      /* :248 */
	}

	if (dry_run > 0) {                                                         /* :251-261 */
		fprintf(stderr, "Dry run completed with %llu clusters. Barcodes found:\n", (unsigned long long)total_reads);
		size_t ne = (size_t)S + extra_barcodes.n;
		oc_entry *ents = (oc_entry *)malloc(sizeof(oc_entry) * (ne ? ne : 1));
		for (int s = 0; s < S; s++) { ents[s].label = samples[s].name.p; ents[s].count = samples[s].total_reads; }
		for (size_t e = 0; e < extra_barcodes.n; e++) { ents[S + e].label = extra_barcodes.ent[e].key.p; ents[S + e].count = extra_barcodes.ent[e].count; }
		print_most_frequent(ents, ne);
	}

	char pct[64];                                                              /* :263-264 */
	oc_fmt_pct(pct, sizeof pct, (double)identified_reads / (double)total_reads * 100.0);
	fprintf(stderr, "%llu / %llu (%s%%) clusters carried a barcode matching one of the provided samples.\n",
	        (unsigned long long)identified_reads, (unsigned long long)total_reads, pct);
	return 0;
}

/* ---- f5: the per-read text commands -------------------------------------------------------- */
static int is_boundary(const oc_str *s, size_t i) { return i == s->n || (i < s->n && (s->p[i] & 0xC0) != 0x80); }
/* &s[a..b] */
static void check_slice(const oc_str *s, size_t a, size_t b)
{
	if (a > b) oc_panic("slice index starts after its end");
	if (b > s->n) oc_panic("byte index out of range of string slice");
	if (!is_boundary(s, a) || !is_boundary(s, b)) oc_panic("byte index is not a char boundary");
}
static void put_str(const oc_str *s) { fwrite(s->p, 1, s->n, stdout); }

static const char *USAGE_TRIMN =
"\nUsage:\n  fasta trim [options] <fastq_file>\n\nOptions:\n"
"  --first=N          Remove first N bases of each read [default: 0].\n"
"  --last=N           Remove last N bases of each read [default: 0].\n";

/* src/fasta_trim.rs:14-48 */
static int trim_fixed(int argc, char **argv)
{
	oc_opt opts[2] = {{"--first", 1, NULL}, {"--last", 1, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 2, opts, 2, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_TRIMN);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :15 */
	uint64_t remove_first, remove_last;
	if (!oc_parse_uint(opts[0].value ? opts[0].value : "0", UINT64_MAX, &remove_first))
		oc_error("N must be a non-negative integer in --first=N.");            /* :16-17 */
	if (!oc_parse_uint(opts[1].value ? opts[1].value : "0", UINT64_MAX, &remove_last))
		oc_error("N must be a non-negative integer in --last=N.");             /* :18-19 */
	oc_str line = {0}, seq = {0}, qual = {0};
	while (oc_read_line(&f, &line)) {                                          /* :24 */
		if (!oc_starts_with(&line, '>') && !oc_starts_with(&line, '@'))
			oc_error("Invalid FASTA/FASTQ format encountered.");               /* :25-27 */
		oc_read_line(&f, &seq);                                                /* :29 */
		uint64_t seq_len = orc_trim_end_len(seq.p, seq.n);                     /* :30 */
		int keep = remove_first + remove_last < seq_len;                       /* :31 usize wraps in a release build */
		if (keep) {                                                            /* :32 */
			check_slice(&seq, remove_first, seq_len - remove_last);
			put_str(&line);
			fwrite(seq.p + remove_first, 1, seq_len - remove_last - remove_first, stdout);
			fputc('\n', stdout);
		} else {                                                               /* :34 */
			put_str(&line);
			fputc('\n', stdout);
		}
		if (oc_starts_with(&line, '@')) {                                      /* :37 */
			oc_read_line(&f, &line);
			oc_read_line(&f, &qual);
			if (keep) {                                                        /* :41 */
				check_slice(&qual, remove_first, seq_len - remove_last);
				fputs("+\n", stdout);
				fwrite(qual.p + remove_first, 1, seq_len - remove_last - remove_first, stdout);
				fputc('\n', stdout);
			} else fputs("+\n\n", stdout);                                     /* :43 */
		}
	}
	return 0;
}

static const char *USAGE_UMI =
"\nUsage:\n  fasta extract dual umi [options] <interleaved_fastq>\n\nOptions:\n"
"  --first-bases=N   First N bases of read contain UMI bases [default: 0]\n";

/* src/fasta_extract_dual_umi.rs:14-72 */
static int extract_dual_umi(int argc, char **argv)
{
	oc_opt opts[1] = {{"--first-bases", 1, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 4, opts, 1, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_UMI);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :16 */
	uint64_t first_bases;
	if (!oc_parse_uint(opts[0].value ? opts[0].value : "0", UINT64_MAX, &first_bases))
		oc_error("N must be a non-negative integer in --first-bases=N.");      /* :17-19 */
	oc_str header_1 = {0}, header_2 = {0}, seq_1 = {0}, seq_2 = {0}, qual_1 = {0}, qual_2 = {0}, line = {0}, umi = {0};
	while (oc_read_line(&f, &header_1)) {                                      /* :30 */
		oc_clear(&umi);
		int fastq_format = 0;
		if (oc_starts_with(&header_1, '@')) fastq_format = 1;                  /* :33-35 */
		else if (oc_starts_with(&header_1, '>')) fastq_format = 0;
		else oc_error("Header is not valid FASTA/FASTQ:\n%s", (const char *)header_1.p);
		if (fastq_format) {                                                    /* :37-47 */
			oc_read_line(&f, &seq_1); oc_read_line(&f, &line); oc_read_line(&f, &qual_1);
			oc_read_line(&f, &header_2); oc_read_line(&f, &seq_2); oc_read_line(&f, &line); oc_read_line(&f, &qual_2);
			if (!oc_starts_with(&header_2, '@')) oc_error("Invalid FASTQ record found in input file.");
		} else {                                                               /* :48-55 */
			oc_read_line(&f, &seq_1); oc_read_line(&f, &header_2); oc_read_line(&f, &seq_2);
			if (!oc_starts_with(&header_2, '>')) oc_error("Invalid FASTA record found in input file.");
		}
		check_slice(&seq_1, 0, first_bases);                                   /* :57 */
		oc_append(&umi, seq_1.p, first_bases);
		oc_append(&umi, "+", 1);                                               /* :58 */
		check_slice(&seq_2, 0, first_bases);                                   /* :59 */
		oc_append(&umi, seq_2.p, first_bases);
		/* :61-70 — all arguments of print! are evaluated before it writes */
		check_slice(&seq_1, first_bases, seq_1.n);
		if (fastq_format) check_slice(&qual_1, first_bases, qual_1.n);
		check_slice(&seq_2, first_bases, seq_2.n);
		if (fastq_format) check_slice(&qual_2, first_bases, qual_2.n);
		const oc_str *hs[2] = {&header_1, &header_2}, *ss[2] = {&seq_1, &seq_2}, *qs[2] = {&qual_1, &qual_2};
		for (int k = 0; k < 2; k++) {
			fwrite(hs[k]->p, 1, orc_trim_end_len(hs[k]->p, hs[k]->n), stdout);
			fputs(" RX:", stdout); put_str(&umi); fputc('\n', stdout);
			fwrite(ss[k]->p + first_bases, 1, ss[k]->n - first_bases, stdout);
			if (fastq_format) { fputs("+\n", stdout); fwrite(qs[k]->p + first_bases, 1, qs[k]->n - first_bases, stdout); }
		}
	}
	return 0;
}

static const char *USAGE_BASESPACE =
"\nUsage:\n  fasta convert basespace <fastq_file>\n\nDescription:\n"
"FASTQ files from Illumina Basespace typically display adapter barcodes at\n"
"the end of the FASTQ header, and have read identifiers that end in /1 or /2.\n"
"This tool replaces the read identifiers by simple consecutive integers, and\n"
"places a \"BC:\" prefix in front of the barcode. An example FASTQ header in\n"
"the output could look like this: @412435 BC:TAGCTACT\n";

/* src/fasta_convert_basespace.rs:17-47 */
static int convert_basespace(int argc, char **argv)
{
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 3, NULL, 0, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_BASESPACE);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :19 */
	uint64_t num_read_pairs = 0;
	oc_str header = {0}, line = {0};
	while (oc_read_line(&f, &header)) {                                        /* :25 */
		num_read_pairs += 1;
		printf("@%llu", (unsigned long long)num_read_pairs);                   /* :27 */
		size_t end = orc_trim_end_len(header.p, header.n);                     /* :32 trim_end().split(':').last() */
		size_t from = end;
		while (from > 0 && header.p[from - 1] != ':') from--;
		if (end > from) { fputs(" BC:", stdout); fwrite(header.p + from, 1, end - from, stdout); }    /* :33 */
		fputc('\n', stdout);                                                   /* :34 */
		if (oc_starts_with(&header, '@')) {                                    /* :36-39 */
			for (int k = 0; k < 3; k++) { oc_read_line(&f, &line); put_str(&line); }
		} else if (oc_starts_with(&header, '>')) {                             /* :40-41 */
			oc_read_line(&f, &line); put_str(&line);
		} else oc_error("Invalid FASTQ line:\n%s", (const char *)header.p);    /* :42-44 */
	}
	return 0;
}

static const char *USAGE_SIMPLIFY =
"\nUsage:\n  fasta simplify read ids [options] <fastq_file>\n\nOptions:\n"
"  --alphanumeric     Use letters a-z, A-Z and 0-9 in read identifiers\n"
"  --discard-umi      Remove \"UMI:\" tags from read identifiers, if present\n";

/* src/fasta_simplify_read_ids.rs:19-62 */
static int simplify_read_ids(int argc, char **argv)
{
	oc_opt opts[2] = {{"--alphanumeric", 0, NULL}, {"--discard-umi", 0, NULL}};
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 4, opts, 2, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_SIMPLIFY);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :21 */
	int discard_umi = opts[1].value != NULL;                                   /* :23 */
	uint64_t read_num = 0;
	oc_str line = {0};
	while (oc_read_line(&f, &line)) {                                          /* :30 */
		if (line.n == 0) continue;                                             /* :31 */
		uint8_t prefix = line.p[0];                                            /* :33 chars().nth(0): a multi-byte char is neither */
		if (prefix != '@' && prefix != '>') oc_error("Invalid FASTA/FASTQ format encountered.");   /* :34-36 */
		read_num += 1;
		printf("%c%llu", prefix, (unsigned long long)read_num);                /* :39 */
		size_t st, en;
		if (!discard_umi && orc_find_umi_field(line.p, line.n, &st, &en)) fwrite(line.p + st, 1, en - st, stdout);   /* :42-46 */
		fputc('\n', stdout);                                                   /* :47 */
		oc_read_line(&f, &line); put_str(&line);                               /* :50-51 */
		if (prefix == '@') {                                                   /* :54-59 */
			oc_read_line(&f, &line);
			fputs("+\n", stdout);
			oc_read_line(&f, &line); put_str(&line);
		}
	}
	return 0;
}

static const char *USAGE_INTERLEAVE = "\nUsage:\n  fasta interleave <fastq_1> <fastq_2>\n";

/* src/fasta_interleave.rs:9-35 */
static int interleave(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 2, NULL, 0, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_INTERLEAVE);
	oc_reader f1 = oc_reader_open(pos[0]);                                     /* :11 */
	oc_reader f2 = oc_reader_open(pos[1]);                                     /* :12 */
	oc_str line = {0};
	while (oc_read_line(&f1, &line)) {                                         /* :15 */
		int lines = 0;
		if (oc_starts_with(&line, '@')) lines = 4;                             /* :16-18 */
		else if (oc_starts_with(&line, '>')) lines = 2;
		else oc_error("Line is not FASTA/FASTQ format: %s", (const char *)line.p);
		put_str(&line);                                                        /* :19 */
		for (int k = 0; k < lines - 1; k++) { oc_read_line(&f1, &line); put_str(&line); }   /* :20-22 */
		oc_read_line(&f2, &line);                                              /* :24 */
		if ((lines == 4 && !oc_starts_with(&line, '@')) || (lines == 2 && !oc_starts_with(&line, '>')))
			oc_error("Input files do not share a consistent format.");         /* :25-28 */
		put_str(&line);                                                        /* :29 */
		for (int k = 0; k < lines - 1; k++) { oc_read_line(&f2, &line); put_str(&line); }   /* :30-32 */
	}
	return 0;
}

static const char *USAGE_CHECK =
"\nUsage:\n  fasta check <fasta/fastq>\n\nDescription:\n"
"Checks that the input FASTA or FASTQ file is correctly formatted, and reports\n"
"the line number if any malformatted lines are found.\n";

/* ReaderWithMemory, src/fasta_check.rs:14-45: the last 10 lines and a line counter */
typedef struct { oc_reader file; uint64_t lines_read; oc_str prev[10]; int nprev; } mem_reader;

static int mem_read_line(mem_reader *r, oc_str *line)
{
	if (!oc_read_line(&r->file, line)) return 0;
	if (r->nprev == 10) {
		oc_str first = r->prev[0];
		memmove(&r->prev[0], &r->prev[1], 9 * sizeof(oc_str));
		r->prev[9] = first;
		r->nprev = 9;
	}
	oc_assign(&r->prev[r->nprev++], line->p, line->n);
	r->lines_read += 1;
	return 1;
}
static const char *mem_history(mem_reader *r)
{
	static oc_str h;
	oc_clear(&h);
	oc_append(&h, "", 0);
	for (int i = 0; i < r->nprev; i++) { oc_append(&h, r->prev[i].p, r->prev[i].n); oc_append(&h, "\n", 1); }
	return (const char *)h.p;
}

/* src/fasta_check.rs:48-70 */
static int check(int argc, char **argv)
{
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 2, NULL, 0, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_CHECK);
	static mem_reader fasta;
	fasta.file = oc_reader_open(pos[0]);
	oc_str line = {0};
	while (mem_read_line(&fasta, &line)) {
		if (oc_starts_with(&line, '>')) {
			mem_read_line(&fasta, &line);
		} else if (oc_starts_with(&line, '@')) {
			mem_read_line(&fasta, &line);
			mem_read_line(&fasta, &line);
			if (!oc_starts_with(&line, '+'))
				oc_error("Missing quality header prefix '+' on line %llu:\n%s\n", (unsigned long long)fasta.lines_read, mem_history(&fasta));
			mem_read_line(&fasta, &line);
		} else {
			oc_error("Missing header prefix '>' or '@' on line %llu:\n%s\n", (unsigned long long)fasta.lines_read, mem_history(&fasta));
		}
	}
	return 0;
}

static const char *USAGE_TO_RAW = "\nUsage:\n  fasta to raw <fasta_file>\n";

/* src/fasta_to_raw.rs:9-29 */
static int to_raw(int argc, char **argv)
{
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 3, NULL, 0, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_TO_RAW);
	oc_reader f = oc_reader_open(pos[0]);
	oc_str line = {0};
	while (oc_read_line(&f, &line)) {
		if (oc_starts_with(&line, '>')) {
			oc_read_line(&f, &line); put_str(&line);
		} else if (oc_starts_with(&line, '@')) {
			oc_read_line(&f, &line); put_str(&line);
			oc_read_line(&f, &line);
			oc_read_line(&f, &line);
		} else oc_error("Invalid FASTA/FASTQ format encountered.");
	}
	return 0;
}

static const char *USAGE_ADD_BASEQ =
"\nUsage:\n  fasta add base qualities <fasta> <baseq>\n\n"
"Converts a FASTA file into a FASTQ file based on user-specified dummy base\nquality values.\n";

/* src/fasta_add_base_qualities.rs:12-31 */
static int add_base_qualities(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 4, NULL, 0, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_ADD_BASEQ);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :14 */
	uint64_t baseq;
	if (!oc_parse_uint(pos[1], 255, &baseq)) oc_error("Base quality must be between 0 - 255.");   /* :15-16 */
	oc_str line = {0};
	while (oc_read_line(&f, &line)) {
		if (oc_starts_with(&line, '>')) {
			fputc('@', stdout); fwrite(line.p + 1, 1, line.n - 1, stdout);      /* :21 */
			oc_read_line(&f, &line);                                           /* :22 */
			size_t seq_len = line.n - 1;                                       /* :23 wraps when the line is empty */
			put_str(&line);                                                    /* :24 */
			if (seq_len == (size_t)-1) oc_panic("capacity overflow");          /* :25 vec!(..; usize::MAX) */
			uint8_t q = (uint8_t)(33 + baseq);                                 /* u8 addition wraps (release build) */
			if (q >= 0x80 && seq_len > 0) oc_panic("called `Result::unwrap()` on an `Err` value: Utf8Error");
			fputs("+\n", stdout);
			for (size_t k = 0; k < seq_len; k++) fputc(q, stdout);
			fputc('\n', stdout);
		} else oc_error("Invalid FASTA format encountered.");                  /* :27-29 */
	}
	return 0;
}

static const char *USAGE_REMOVE_BASEQ = "\nUsage:\n  fasta remove base qualities <fastq_file>\n";

/* src/fasta_remove_base_qualities.rs:9-27 */
static int remove_base_qualities(int argc, char **argv)
{
	const char *pos[1]; int npos;
	if (!oc_parse(argc, argv, 4, NULL, 0, pos, &npos, 1) || npos != 1) oc_error("Invalid arguments.\n%s", USAGE_REMOVE_BASEQ);
	oc_reader f = oc_reader_open(pos[0]);
	oc_str line = {0};
	while (oc_read_line(&f, &line)) {
		if (oc_starts_with(&line, '@')) {
			fputc('>', stdout); fwrite(line.p + 1, 1, line.n - 1, stdout);      /* :16 */
			oc_read_line(&f, &line); put_str(&line);                           /* :17-18 */
			oc_read_line(&f, &line);                                           /* :20-21 */
			oc_read_line(&f, &line);
		} else oc_error("Invalid FASTQ format encountered.");                  /* :23-25 */
	}
	return 0;
}

static const char *USAGE_DEINTERLEAVE = "\nUsage:\n  fasta deinterleave <interleaved_fastq> <out_prefix>\n";

/* src/fasta_deinterleave.rs:9-39 */
static int deinterleave(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 2, NULL, 0, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_DEINTERLEAVE);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :11 */
	char path[4096];
	snprintf(path, sizeof path, "%s_1.fq.gz", pos[1]);
	FILE *out_1 = oc_gzip_writer(path, 0);                                     /* :13-14 */
	snprintf(path, sizeof path, "%s_2.fq.gz", pos[1]);
	FILE *out_2 = oc_gzip_writer(path, 0);                                     /* :15-16 */
	oc_str line = {0};
	while (oc_read_line(&f, &line)) {                                          /* :19 */
		int lines = 0;
		if (oc_starts_with(&line, '@')) lines = 4;                             /* :20-22 */
		else if (oc_starts_with(&line, '>')) lines = 2;
		else oc_error("Line is not FASTA/FASTQ format: %s", (const char *)line.p);
		fwrite(line.p, 1, line.n, out_1);                                      /* :23 */
		for (int k = 0; k < lines - 1; k++) { oc_read_line(&f, &line); fwrite(line.p, 1, line.n, out_1); }   /* :24-26 */
		oc_read_line(&f, &line);                                               /* :28 */
		if ((lines == 4 && !oc_starts_with(&line, '@')) || (lines == 2 && !oc_starts_with(&line, '>')))
			oc_error("Interleaved FASTA records are not in consistent format.");       /* :29-32 */
		fwrite(line.p, 1, line.n, out_2);                                      /* :33 */
		for (int k = 0; k < lines - 1; k++) { oc_read_line(&f, &line); fwrite(line.p, 1, line.n, out_2); }   /* :34-36 */
	}
	return 0;
}

static const char *USAGE_ANCHORS = "\nUsage:\n  fasta split into anchors <fastq> <anchor_len>\n";

/* src/fasta_split_into_anchors.rs:10-45 */
static int split_into_anchors(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 4, NULL, 0, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_ANCHORS);
	oc_reader f = oc_reader_open(pos[0]);                                      /* :12 */
	uint64_t anchor_len;
	if (!oc_parse_uint(pos[1], UINT64_MAX, &anchor_len)) oc_error("<anchor_len> must be a positive integer.");   /* :13-14 */
	uint64_t reads = 0;
	oc_str header = {0}, seq = {0}, qual = {0}, line = {0};
	while (oc_read_line(&f, &header)) {                                        /* :22 */
		reads += 1;
		oc_read_line(&f, &seq);                                                /* :25 */
		uint64_t seq_len = orc_trim_end_len(seq.p, seq.n);                     /* :26 */
		if (seq_len < anchor_len * 2) continue;                                /* :27 — the '+' and quality lines of a FASTQ record stay unread */
		if (oc_starts_with(&header, '@')) {                                    /* :29-36 */
			oc_read_line(&f, &line);
			oc_read_line(&f, &qual);
			check_slice(&seq, 0, anchor_len); check_slice(&qual, 0, anchor_len);
			printf("@%llu\n", (unsigned long long)reads);
			fwrite(seq.p, 1, anchor_len, stdout); fputs("\n+\n", stdout); fwrite(qual.p, 1, anchor_len, stdout); fputc('\n', stdout);
			check_slice(&seq, seq_len - anchor_len, seq_len); check_slice(&qual, seq_len - anchor_len, seq_len);
			printf("@%llu\n", (unsigned long long)reads);
			fwrite(seq.p + seq_len - anchor_len, 1, anchor_len, stdout); fputs("\n+\n", stdout);
			fwrite(qual.p + seq_len - anchor_len, 1, anchor_len, stdout); fputc('\n', stdout);
		} else if (oc_starts_with(&header, '>')) {                             /* :37-39 */
			check_slice(&seq, 0, anchor_len);
			printf(">%llu\n", (unsigned long long)reads);
			fwrite(seq.p, 1, anchor_len, stdout); fputc('\n', stdout);
			check_slice(&seq, seq_len - anchor_len, seq_len);
			printf(">%llu\n", (unsigned long long)reads);
			fwrite(seq.p + seq_len - anchor_len, 1, anchor_len, stdout); fputc('\n', stdout);
		} else oc_error("Header is not valid FASTA/FASTQ:\n%s", (const char *)header.p);   /* :40-42 */
	}
	return 0;
}

static const char *USAGE_GC =
"\nUsage:\n  fasta gc content <genome.fa> <regions.bed>\n\nDescription:\n"
"Calculates the GC content percentage of FASTA file regions listed in the input\n"
"BED file. Ambiguous N nucleotides are omitted from both the numerator and the\n"
"denominator.\n";

/* src/fasta_gc_content.rs:17-50.  The FASTA reader is rust-bio 0.19's (not in the tree): restated from its published
 * behaviour — a record is a '>' line plus the lines up to the next '>' line, each trim_right()ed; id = header[1..]
 * trimmed, up to its first space. */
typedef struct { oc_str id, seq; } chrom_t;

static int gc_content(int argc, char **argv)
{
	const char *pos[2]; int npos;
	if (!oc_parse(argc, argv, 3, NULL, 0, pos, &npos, 2) || npos != 2) oc_error("Invalid arguments.\n%s", USAGE_GC);
	fputs("Reading reference genome into memory...\n", stderr);                 /* :22 */
	FILE *fa = fopen(pos[0], "rb");                                            /* :23 */
	if (!fa) oc_error("Input FASTA file %s could not be read.", pos[0]);
	chrom_t *chr = NULL; size_t nchr = 0, capchr = 0;
	{
		oc_str line = {0};
		int have = 0;
		for (;;) {                                                             /* fasta::Reader::read, one record */
			if (!have) {
				/* read_line's Err (invalid UTF-8) is unwrap()ed at :27: a panic, not error! */
				oc_clear(&line);
				int c;
				while ((c = getc(fa)) != EOF) { uint8_t b = (uint8_t)c; oc_append(&line, &b, 1); if (c == '\n') break; }
				if (line.n == 0) break;
				if (!orc_utf8_valid(line.p, line.n)) oc_panic("called `Result::unwrap()` on an `Err` value: stream did not contain valid UTF-8");
			}
			if (line.p[0] != '>') oc_panic("called `Result::unwrap()` on an `Err` value: Expected > at record start.");
			size_t hend = orc_trim_end_len(line.p, line.n), e = 1;
			while (e < hend && line.p[e] != ' ') e++;
			if (nchr == capchr) { capchr = capchr ? capchr * 2 : 64; chr = (chrom_t *)realloc(chr, capchr * sizeof *chr); }
			size_t slot = nchr;
			for (size_t k = 0; k < nchr; k++)                                   /* HashMap::insert: a repeated id replaces */
				if (chr[k].id.n == e - 1 && memcmp(chr[k].id.p, line.p + 1, e - 1) == 0) { slot = k; break; }
			if (slot == nchr) { memset(&chr[nchr], 0, sizeof(chrom_t)); oc_assign(&chr[nchr].id, line.p + 1, e - 1); nchr++; }
			oc_clear(&chr[slot].seq);
			oc_append(&chr[slot].seq, "", 0);
			for (;;) {
				oc_clear(&line);
				int c;
				while ((c = getc(fa)) != EOF) { uint8_t b = (uint8_t)c; oc_append(&line, &b, 1); if (c == '\n') break; }
				if (line.n && !orc_utf8_valid(line.p, line.n)) oc_panic("called `Result::unwrap()` on an `Err` value: stream did not contain valid UTF-8");
				if (line.n == 0 || line.p[0] == '>') break;
				oc_append(&chr[slot].seq, line.p, orc_trim_end_len(line.p, line.n));
			}
			have = line.n > 0;
			if (!have) break;
		}
		fclose(fa);
	}
	oc_reader bed = oc_reader_open(pos[1]);                                     /* :31 */
	oc_str line = {0};
	while (oc_read_line(&bed, &line)) {                                        /* :33 */
		size_t off = orc_trim_start_off(line.p, line.n), end = orc_trim_end_len(line.p, line.n);
		if (end < off) end = off;
		size_t cs[3] = {0, 0, 0}, ce[3] = {0, 0, 0}; int ncol = 0;             /* :34 split('\t') */
		for (size_t a = off;;) {
			size_t b = a;
			while (b < end && line.p[b] != '\t') b++;
			if (ncol < 3) { cs[ncol] = a; ce[ncol] = b; }
			ncol++;
			if (b >= end) break;
			a = b + 1;
		}
		if (ncol < 3) fprintf(stderr, "WARNING: Input BED file contains line with less than 3 columns:\n%s\n\n", (const char *)line.p);   /* :35-37 */
		chrom_t *c = NULL;                                                     /* :39 */
		for (size_t k = 0; k < nchr; k++)
			if (chr[k].id.n == ce[0] - cs[0] && memcmp(chr[k].id.p, line.p + cs[0], chr[k].id.n) == 0) { c = &chr[k]; break; }
		if (!c) continue;
		char num[64];
		uint64_t start, stop;
		if (ncol < 2) oc_panic("index out of bounds: the len is 1 but the index is 1");
		size_t l1 = ce[1] - cs[1];
		if (l1 >= sizeof num) oc_error("Invalid region:\n%s\n", (const char *)line.p);
		memcpy(num, line.p + cs[1], l1); num[l1] = 0;
		if (!oc_parse_uint(num, UINT64_MAX, &start)) oc_error("Invalid region:\n%s\n", (const char *)line.p);      /* :40 */
		if (ncol < 3) oc_panic("index out of bounds: the len is 2 but the index is 2");
		size_t l2 = ce[2] - cs[2];
		if (l2 >= sizeof num) oc_error("Invalid region:\n%s\n", (const char *)line.p);
		memcpy(num, line.p + cs[2], l2); num[l2] = 0;
		if (!oc_parse_uint(num, UINT64_MAX, &stop)) oc_error("Invalid region:\n%s\n", (const char *)line.p);       /* :41 */
		if (start > stop || stop > c->seq.n) oc_error("Invalid region:\n%s\n", (const char *)line.p);             /* :42 */
		uint64_t gc, total;
		orc_gc_count(c->seq.p + start, (size_t)(stop - start), &gc, &total);   /* :44-45 */
		float ratio = (float)gc / (float)total;                                /* :46 */
		if (total == 0) printf("%llu\t%llu\tNaN\n", (unsigned long long)gc, (unsigned long long)total);
		else printf("%llu\t%llu\t%.3f\n", (unsigned long long)gc, (unsigned long long)total, (double)ratio);
	}
	return 0;
}

/* src/fasta_main.rs:42-82 */
int main(int argc, char **argv)
{
	int rc;
	if (argc >= 2 && !strcmp(argv[1], "check"))
		rc = check(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "to") && !strcmp(argv[2], "raw"))
		rc = to_raw(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "add") && !strcmp(argv[2], "base") && !strcmp(argv[3], "qualities"))
		rc = add_base_qualities(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "remove") && !strcmp(argv[2], "base") && !strcmp(argv[3], "qualities"))
		rc = remove_base_qualities(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "simplify") && !strcmp(argv[2], "read") && !strcmp(argv[3], "ids"))
		rc = simplify_read_ids(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "interleave"))
		rc = interleave(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "deinterleave"))
		rc = deinterleave(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "split") && !strcmp(argv[2], "into") && !strcmp(argv[3], "anchors"))
		rc = split_into_anchors(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "trim") && !strcmp(argv[2], "by") && !strcmp(argv[3], "quality"))
		rc = trim_by_quality(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "trim"))
		rc = trim_fixed(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "mask") && !strcmp(argv[2], "by") && !strcmp(argv[3], "quality"))
		rc = mask_by_quality(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "gc") && !strcmp(argv[2], "content"))
		rc = gc_content(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "add") && !strcmp(argv[2], "barcode"))
		rc = add_barcode(argc, argv);
	else if (argc >= 4 && !strcmp(argv[1], "extract") && !strcmp(argv[2], "dual") && !strcmp(argv[3], "umi"))
		rc = extract_dual_umi(argc, argv);
	else if (argc >= 3 && !strcmp(argv[1], "convert") && !strcmp(argv[2], "basespace"))
		rc = convert_basespace(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "demultiplex"))
		rc = demultiplex(argc, argv);
	else if (argc >= 2 && !strcmp(argv[1], "statistics"))
		rc = statistics(argc, argv);
	else {
		fprintf(stderr, "%s\n", USAGE_TOP);
		rc = 0;
	}
	fflush(stdout);
	oc_wait_children();
	return rc;
}
