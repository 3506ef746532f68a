/*
 * seqkit_oracle.c — CPU ORACLE (test infrastructure; see seqkit_oracle.h header note).
 * PARITY UNPINNED against the reference binary; pinned only to the hand-derived
 * known-answer vectors of SURVEY.md Appendix A (tests/golden/).
 *
 * Every function follows the cited reference lines one statement at a time; scalar,
 * single-threaded, no SIMD tricks, so that it reads like the source it restates.
 */
#include "seqkit_oracle.h"

#include <stdlib.h>
#include <string.h>

/* src/fasta_trim_by_quality.rs:28-42
 *   total = lowest_total = -50; k = lowest_k = trim_end(qual).len()
 *   while k > 0 { k -= 1; total += (qual[k] - 33u8) as i32 - min_baseq as i32;
 *                 if total > 0 {break}; if total < lowest_total {lowest_total = total; lowest_k = k} }
 * `qual[k] - 33u8` is u8 arithmetic: wraps in release builds.                        */
uint32_t orc_trim_lowest_k(const uint8_t *qual, uint32_t n, uint8_t min_baseq)
{
	int32_t total = -50;
	int32_t lowest_total = total;
	uint32_t k = n;
	uint32_t lowest_k = k;
	while (k > 0) {
		k -= 1;
		uint8_t v = (uint8_t)(qual[k] - (uint8_t)33);
		total += (int32_t)v - (int32_t)min_baseq;
		if (total > 0) break;
		if (total < lowest_total) {
			lowest_total = total;
			lowest_k = k;
		}
	}
	return lowest_k;
}

/* src/fasta_mask_by_quality.rs:40-43
 *   output.push(if qual as u8 - 33u8 < min_baseq { 'N' } else { base })               */
void orc_mask_bytes(uint8_t *seq, const uint8_t *qual, uint32_t n, uint8_t min_baseq)
{
	for (uint32_t i = 0; i < n; i++) {
		uint8_t v = (uint8_t)(qual[i] - (uint8_t)33);
		if (v < min_baseq) seq[i] = 'N';
	}
}

/* src/fasta_demultiplex.rs:269-277 */
size_t orc_barcode_diff(const uint8_t *observed, const uint8_t *candidate, size_t len)
{
	size_t mismatches = 0;
	for (size_t k = 0; k < len; k++) {
		if (candidate[k] == 'N' || candidate[k] == 'U') continue;
		if (observed[k] != candidate[k]) mismatches += 1;
	}
	return mismatches;
}

/* src/fasta_demultiplex.rs:154-166 */
orc_match orc_best_match(const uint8_t *observed, const uint8_t *table, int S, int L)
{
	orc_match m;
	m.best_sample = 0;
	m.equally_fine = 0;
	m.lowest_diff = UINT64_MAX;
	for (int s = 0; s < S; s++) {
		uint64_t diff = orc_barcode_diff(observed, table + (size_t)s * (size_t)L, (size_t)L);
		if (diff < m.lowest_diff) {
			m.lowest_diff = diff;
			m.best_sample = (uint32_t)s;
			m.equally_fine = (uint32_t)s;
		} else if (diff == m.lowest_diff) {
			m.equally_fine = (uint32_t)s;
		}
	}
	return m;
}

/* src/fasta_demultiplex.rs:168-189 */
int32_t orc_decide(orc_match m, uint64_t max_diff)
{
	if (m.lowest_diff <= max_diff) {
		if (m.best_sample == m.equally_fine) return (int32_t)m.best_sample;
		return ORC_AMBIGUOUS;
	}
	return ORC_NONE;
}

void orc_trim_batch(const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                    uint8_t min_baseq, uint16_t *lowest_k)
{
	for (int64_t r = 0; r < n; r++) {
		uint32_t l = len ? len[r] : (uint32_t)stride;
		lowest_k[r] = (uint16_t)orc_trim_lowest_k(qual + r * (int64_t)stride, l, min_baseq);
	}
}

void orc_mask_batch(uint8_t *seq, const uint8_t *qual, const uint16_t *len, int stride,
                    int64_t n, uint8_t min_baseq)
{
	for (int64_t r = 0; r < n; r++) {
		uint32_t l = len ? len[r] : (uint32_t)stride;
		orc_mask_bytes(seq + r * (int64_t)stride, qual + r * (int64_t)stride, l, min_baseq);
	}
}

void orc_demux_batch(const uint8_t *table, int S, int L, int max_diff,
                     const uint8_t *bc, int bc_stride, int64_t n,
                     int32_t *assign, uint8_t *lowest_diff, int16_t *first_idx,
                     int16_t *last_idx, uint64_t *counts)
{
	for (int64_t r = 0; r < n; r++) {
		orc_match m = orc_best_match(bc + r * (int64_t)bc_stride, table, S, L);
		int32_t code = orc_decide(m, (uint64_t)max_diff);
		assign[r] = code;
		if (lowest_diff) lowest_diff[r] = m.lowest_diff > 255 ? 255 : (uint8_t)m.lowest_diff;
		if (first_idx) first_idx[r] = (int16_t)m.best_sample;
		if (last_idx) last_idx[r] = (int16_t)m.equally_fine;
		if (counts) {
			counts[S] += 1;                                   /* total_reads += 1  :169 */
			if (code >= 0) { counts[S + 1] += 1; counts[code] += 1; }   /* :177-178 */
			else if (code == ORC_AMBIGUOUS) counts[S + 2] += 1;
		}
	}
}

/* src/sam_statistics.rs:63-69 and src/sam_fragment_lengths.rs:29-43.
 * Flag predicates are the SAM-spec bits that rust-htslib 0.31's Record::is_* test. */
#define F_PAIRED 0x1
#define F_UNMAPPED 0x4
#define F_MUNMAP 0x8
#define F_FIRST 0x40
#define F_SECONDARY 0x100
#define F_DUP 0x400
#define F_SUPPL 0x800

static int frag_keep(uint16_t f, int32_t tid, int32_t mtid, int32_t tlen, int32_t max_frag,
                     uint64_t *frag)
{
	if (!(f & F_PAIRED)) return 0;
	if (!(f & F_FIRST)) return 0;
	if ((f & F_UNMAPPED) || (f & F_MUNMAP)) return 0;
	if ((f & F_DUP) || (f & F_SECONDARY)) return 0;
	if (f & F_SUPPL) return 0;
	if (tid != mtid) return 0;
	/* insert_size().abs() as usize: i64 abs of a value that was i32 on disk */
	int64_t t = (int64_t)tlen;
	uint64_t a = (uint64_t)(t < 0 ? -t : t);
	if (a > (uint64_t)(int64_t)max_frag) return 0;
	*frag = a;
	return 1;
}

void orc_bam_flag_tlen(const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                       const int32_t *tlen, int64_t n, int32_t max_frag,
                       uint64_t counters[3], uint64_t *hist, uint64_t *hist_total)
{
	for (int64_t i = 0; i < n; i++) {
		uint16_t f = flag[i];
		if (counters) {
			if (!((f & F_SECONDARY) || (f & F_SUPPL))) {
				counters[0] += 1;
				if (!(f & F_UNMAPPED)) {
					counters[1] += 1;
					if (f & F_DUP) counters[2] += 1;
				}
			}
		}
		if (hist) {
			uint64_t a;
			if (frag_keep(f, tid[i], mtid[i], tlen[i], max_frag, &a)) {
				if (hist_total) *hist_total += 1;
				hist[a] += 1;
			}
		}
	}
}

int64_t orc_fragment_lengths_stop(const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                                  const int32_t *tlen, int64_t n, int32_t max_frag,
                                  uint64_t stop_after, uint64_t *hist, uint64_t *hist_total)
{
	uint64_t total = 0;
	int64_t i = 0;
	for (; i < n; i++) {
		uint64_t a;
		if (!frag_keep(flag[i], tid[i], mtid[i], tlen[i], max_frag, &a)) continue;
		total += 1;
		hist[a] += 1;
		if (total >= stop_after) { i++; break; }
	}
	if (hist_total) *hist_total += total;
	return i;
}

/* src/sam_fragments.rs:27-38 */
int64_t orc_fragments_keep(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                           int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep)
{
	int64_t kept = 0;
	for (int64_t i = 0; i < n; i++) {
		uint16_t f = flag[i];
		keep[i] = 0;
		if (!(f & F_PAIRED)) continue;                              /* :28 */
		if ((f & F_UNMAPPED) || (f & F_MUNMAP)) continue;           /* :29 */
		if ((f & F_DUP) || (f & F_SECONDARY)) continue;             /* :30 */
		if (f & F_SUPPL) continue;                                  /* :31 */
		if (tid[i] != mtid[i]) continue;                            /* :32 */
		if ((f & 0x10) || !(f & 0x20)) continue;                    /* :33 is_reverse || !is_mate_reverse */
		if (f & 0x200) continue;                                    /* :34 quality check failed */
		int64_t t = (int64_t)tlen[i];
		int64_t frag = t < 0 ? -t : t;                              /* :37 */
		if (frag > max_size || frag < min_size) continue;           /* :38 */
		keep[i] = 1;
		kept++;
	}
	return kept;
}

/* ---------------------------------------------------------------------------------- */
/* Rust std text semantics                                                            */

static int is_rust_whitespace(uint32_t c)
{
	if (c >= 0x09 && c <= 0x0D) return 1;
	if (c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680) return 1;
	if (c >= 0x2000 && c <= 0x200A) return 1;
	if (c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000) return 1;
	return 0;
}

/* decode the char that ENDS at s[n) (valid UTF-8 assumed); returns its byte length */
static size_t utf8_prev(const uint8_t *s, size_t n, uint32_t *cp)
{
	size_t i = n - 1;
	while (i > 0 && (s[i] & 0xC0) == 0x80 && n - i < 4) i--;
	size_t l = n - i;
	uint32_t c;
	if (l == 1) c = s[i];
	else if (l == 2) c = ((uint32_t)(s[i] & 0x1F) << 6) | (s[i + 1] & 0x3F);
	else if (l == 3) c = ((uint32_t)(s[i] & 0x0F) << 12) | ((uint32_t)(s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F);
	else c = ((uint32_t)(s[i] & 0x07) << 18) | ((uint32_t)(s[i + 1] & 0x3F) << 12) |
	         ((uint32_t)(s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F);
	*cp = c;
	return l;
}

static size_t utf8_next(const uint8_t *s, size_t n, uint32_t *cp)
{
	uint8_t b = s[0];
	size_t l = b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : 4;
	if (l > n) l = n;
	uint32_t c;
	if (l == 1) c = b;
	else if (l == 2) c = ((uint32_t)(b & 0x1F) << 6) | (s[1] & 0x3F);
	else if (l == 3) c = ((uint32_t)(b & 0x0F) << 12) | ((uint32_t)(s[1] & 0x3F) << 6) | (s[2] & 0x3F);
	else c = ((uint32_t)(b & 0x07) << 18) | ((uint32_t)(s[1] & 0x3F) << 12) |
	         ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
	*cp = c;
	return l;
}

size_t orc_trim_end_len(const uint8_t *s, size_t n)
{
	while (n > 0) {
		uint32_t c;
		size_t l = utf8_prev(s, n, &c);
		if (!is_rust_whitespace(c)) break;
		n -= l;
	}
	return n;
}

size_t orc_trim_start_off(const uint8_t *s, size_t n)
{
	size_t off = 0;
	while (off < n) {
		uint32_t c;
		size_t l = utf8_next(s + off, n - off, &c);
		if (!is_rust_whitespace(c)) break;
		off += l;
	}
	return off;
}

/* regex " UMI:[^\s]*" (src/fasta_simplify_read_ids.rs:26): leftmost " UMI:", then every
 * char that is not Unicode White_Space.                                               */
int orc_find_umi_field(const uint8_t *hdr, size_t n, size_t *start, size_t *end)
{
	for (size_t i = 0; i + 5 <= n; i++) {
		if (memcmp(hdr + i, " UMI:", 5) != 0) continue;
		size_t e = i + 5;
		while (e < n) {
			uint32_t c;
			size_t l = utf8_next(hdr + e, n - e, &c);
			if (is_rust_whitespace(c)) break;
			e += l;
		}
		*start = i;
		*end = e;
		return 1;
	}
	return 0;
}

/* Well-formed UTF-8 per the Unicode standard table 3-7 (what core::str::from_utf8 accepts). */
int orc_utf8_valid(const uint8_t *s, size_t n)
{
	size_t i = 0;
	while (i < n) {
		uint8_t b = s[i];
		if (b < 0x80) { i++; continue; }
		if (b >= 0xC2 && b <= 0xDF) {
			if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return 0;
			i += 2; continue;
		}
		if (b >= 0xE0 && b <= 0xEF) {
			if (i + 2 >= n) return 0;
			uint8_t c1 = s[i + 1], c2 = s[i + 2];
			uint8_t lo = 0x80, hi = 0xBF;
			if (b == 0xE0) lo = 0xA0;
			if (b == 0xED) hi = 0x9F;
			if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80) return 0;
			i += 3; continue;
		}
		if (b >= 0xF0 && b <= 0xF4) {
			if (i + 3 >= n) return 0;
			uint8_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
			uint8_t lo = 0x80, hi = 0xBF;
			if (b == 0xF0) lo = 0x90;
			if (b == 0xF4) hi = 0x8F;
			if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return 0;
			i += 4; continue;
		}
		return 0;
	}
	return 1;
}

static int bc_class(uint8_t c)
{
	switch (c) {
	case 'A': case 'C': case 'G': case 'T': case 'N':
	case 'a': case 'c': case 'g': case 't': case 'n': case '+':
		return 1;
	default:
		return 0;
	}
}

/* regex crate semantics for a literal prefix followed by one greedy class repetition:
 * the leftmost position where " BC:" is followed by >= 1 class byte; the match then
 * extends over the whole run of class bytes.                                         */
int orc_find_bc_field(const uint8_t *hdr, size_t n, size_t *start, size_t *end)
{
	for (size_t i = 0; i + 5 <= n; i++) {
		if (hdr[i] == ' ' && hdr[i + 1] == 'B' && hdr[i + 2] == 'C' && hdr[i + 3] == ':' &&
		    bc_class(hdr[i + 4])) {
			size_t e = i + 5;
			while (e < n && bc_class(hdr[e])) e++;
			*start = i;
			*end = e;
			return 1;
		}
	}
	return 0;
}

static int bc_class_stats(uint8_t c) { return c != '+' && bc_class(c); }

int orc_find_bc_field_stats(const uint8_t *hdr, size_t n, size_t *start, size_t *end)
{
	for (size_t i = 0; i + 5 <= n; i++) {
		if (hdr[i] == ' ' && hdr[i + 1] == 'B' && hdr[i + 2] == 'C' && hdr[i + 3] == ':' &&
		    bc_class_stats(hdr[i + 4])) {
			size_t e = i + 5;
			while (e < n && bc_class_stats(hdr[e])) e++;
			*start = i;
			*end = e;
			return 1;
		}
	}
	return 0;
}

/* f3: string-keyed counting map, entries kept in insertion order */
static uint64_t fnv1a(const char *s)
{
	uint64_t h = 1469598103934665603ull;
	for (; *s; s++) { h ^= (uint8_t)*s; h *= 1099511628211ull; }
	return h;
}

int64_t orc_census(const uint8_t *bc, int stride, int L, int64_t n, const int32_t *assign,
                   int64_t row_base, orc_census_entry **out)
{
	size_t cap = 1024, used = 0, ecap = 256;
	int64_t *slot = malloc(cap * sizeof *slot);
	orc_census_entry *ent = malloc(ecap * sizeof *ent);
	*out = NULL;
	if (!slot || !ent || L < 0 || L > 31) { free(slot); free(ent); return -1; }
	for (size_t i = 0; i < cap; i++) slot[i] = -1;
	for (int64_t r = 0; r < n; r++) {
		if (assign && assign[r] != ORC_NONE) continue;
		char key[32];
		int k = 0;
		const uint8_t *row = bc + r * (int64_t)stride;
		while (k < L && row[k] != 0) { key[k] = (char)row[k]; k++; }
		memset(key + k, 0, sizeof key - (size_t)k);
		size_t i = fnv1a(key) & (cap - 1);
		while (slot[i] >= 0 && memcmp(ent[slot[i]].barcode, key, 32) != 0) i = (i + 1) & (cap - 1);
		if (slot[i] >= 0) { ent[slot[i]].count++; continue; }
		if (used == ecap) {
			ecap *= 2;
			orc_census_entry *ne = realloc(ent, ecap * sizeof *ent);
			if (!ne) { free(slot); free(ent); return -1; }
			ent = ne;
		}
		memcpy(ent[used].barcode, key, 32);
		ent[used].count = 1;
		ent[used].first_row = row_base + r;
		slot[i] = (int64_t)used++;
		if (used * 2 > cap) {                          /* rehash at load 1/2 */
			cap *= 2;
			int64_t *ns = realloc(slot, cap * sizeof *slot);
			if (!ns) { free(slot); free(ent); return -1; }
			slot = ns;
			for (size_t j = 0; j < cap; j++) slot[j] = -1;
			for (size_t e = 0; e < used; e++) {
				size_t j = fnv1a(ent[e].barcode) & (cap - 1);
				while (slot[j] >= 0) j = (j + 1) & (cap - 1);
				slot[j] = (int64_t)e;
			}
		}
	}
	free(slot);
	*out = ent;
	return (int64_t)used;
}

void orc_free(void *p) { free(p); }

/* ---- f4: src/sam_to_fastq.rs:31-59 ------------------------------------------------- */
static uint8_t encoded_base(const uint8_t *seq4, uint32_t k)
{
	return (uint8_t)((seq4[k / 2] >> (4 * (1 - (k & 1)))) & 15);     /* rust-htslib Seq::encoded_base */
}

void orc_bam_sequence(const uint8_t *seq4, const uint8_t *qual, uint32_t l, int reverse,
                      uint8_t min_baseq, uint8_t *out)
{
	uint32_t o = 0;
	if (reverse) {                                                   /* :36-46 */
		for (uint32_t k = l; k-- > 0;) {
			if (qual[k] < min_baseq) out[o++] = 'N';
			else {
				switch (encoded_base(seq4, k)) {
				case 1: out[o++] = 'T'; break;
				case 2: out[o++] = 'G'; break;
				case 4: out[o++] = 'C'; break;
				case 8: out[o++] = 'A'; break;
				default: out[o++] = 'N'; break;
				}
			}
		}
	} else {                                                         /* :47-57 */
		for (uint32_t k = 0; k < l; k++) {
			if (qual[k] < min_baseq) out[o++] = 'N';
			else {
				switch (encoded_base(seq4, k)) {
				case 1: out[o++] = 'A'; break;
				case 2: out[o++] = 'C'; break;
				case 4: out[o++] = 'G'; break;
				case 8: out[o++] = 'T'; break;
				default: out[o++] = 'N'; break;
				}
			}
		}
	}
}

void orc_bam_sequence_batch(const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride,
                            const uint16_t *len, const uint16_t *flag, int64_t n,
                            uint8_t min_baseq, uint8_t *out)
{
	for (int64_t r = 0; r < n; r++)
		orc_bam_sequence(seq4 + r * (int64_t)seq4_stride, qual + r * (int64_t)stride,
		                 len ? len[r] : (uint32_t)stride, (flag[r] & 0x10) != 0, min_baseq,
		                 out + r * (int64_t)stride);
}

/* ---- f2 (second half): src/sam_count.rs ------------------------------------------------ */
int orc_count_interval(uint16_t flag, uint8_t mapq, int32_t tid, int32_t mtid, int32_t pos, int32_t mpos,
                       int32_t tlen, int32_t end_pos, const orc_count_params *p, uint32_t *start_out, uint32_t *end_out)
{
	if (flag & 0x4) return 0;                                          /* :46 is_unmapped */
	if ((flag & 0x400) || (flag & 0x100)) return 0;                    /* :47 duplicate, secondary */
	if (flag & 0x800) return 0;                                        /* :48 supplementary */
	if (mapq < p->min_mapq) return 0;                                  /* :49 */
	uint32_t start = (uint32_t)pos;                                    /* :75 */
	uint32_t end;
	if (p->single_end) {
		end = (uint32_t)end_pos;                                       /* :77 */
	} else {
		if (!(flag & 0x1)) return 0;                                   /* :79 */
		if (flag & 0x8) return 0;                                      /* :80 */
		if (tid != mtid) return 0;                                     /* :81 */
		if (pos > mpos || (pos == mpos && !(flag & 0x40))) return 0;   /* :91 */
		int64_t ins = tlen;                                            /* :93 insert_size().abs() as u32 */
		if (ins < 0) ins = -ins;
		uint32_t insert_size = (uint32_t)ins;
		if (insert_size < 20) return 0;                                /* :94 */
		end = start + insert_size;                                     /* :96 (u32, wraps) */
	}
	if ((uint32_t)(end - start) > p->max_frag_len) return 0;           /* :99 */
	if (p->count_centers) {                                            /* :103-107 */
		uint32_t len = end - start;
		start += len / 2;
		end = start + 1;
	}
	*start_out = start;
	*end_out = end;
	return 1;
}

void orc_count_overlaps(const uint32_t *rstart, const uint32_t *rend, const int64_t *idx, int64_t nreg,
                        uint32_t start, uint32_t end, uint32_t *region_frags)
{
	for (int64_t i = 0; i < nreg; i++) {                               /* :122-126 */
		if (rstart[idx[i]] >= end) break;
		if (rend[idx[i]] <= start) continue;
		region_frags[idx[i]] += 1;
	}
}

void orc_count_state_init(orc_count_state *st) { memset(st, 0, sizeof *st); st->prev_chr = -1; }
void orc_count_state_free(orc_count_state *st) { free(st->deque); st->deque = NULL; }

int orc_count_record(orc_count_state *st, uint16_t flag, uint8_t mapq, int32_t tid, int32_t mtid, int32_t pos,
                     int32_t mpos, int32_t tlen, int32_t end_pos, const orc_count_params *p, int32_t n_chr,
                     const int32_t *rchr, const uint32_t *rstart, const uint32_t *rend, int64_t n_regions,
                     uint32_t *region_frags)
{
	if (flag & 0x4) return 0;                                          /* :46 */
	if ((flag & 0x400) || (flag & 0x100)) return 0;                    /* :47 */
	if (flag & 0x800) return 0;                                        /* :48 */
	if (mapq < p->min_mapq) return 0;                                  /* :49 */
	if (tid != st->prev_chr) {                                         /* :52-67 */
		st->prev_chr = tid;
		if (tid < 0 || tid >= n_chr) return ORC_COUNT_BAD_TID;         /* :55 chr_names[read.tid() as usize] */
		if (st->cap < n_regions) {
			st->cap = n_regions;
			st->deque = (int64_t *)realloc(st->deque, (size_t)(n_regions ? n_regions : 1) * sizeof(int64_t));
		}
		st->front = 0; st->len = 0;
		for (int64_t r = 0; r < n_regions; r++)                        /* :61-63 */
			if (rchr[r] == tid) st->deque[st->len++] = r;
		for (int64_t a = 1; a < st->len; a++) {                        /* :64 sort_by_key(start): stable */
			int64_t x = st->deque[a], b = a;
			while (b > 0 && rstart[st->deque[b - 1]] > rstart[x]) { st->deque[b] = st->deque[b - 1]; b--; }
			st->deque[b] = x;
		}
	} else if ((int64_t)pos < st->prev_pos) {                          /* :69-72 */
		return ORC_COUNT_UNSORTED;
	}
	st->prev_pos = pos;                                                /* :73 */
	uint32_t start, end;                                               /* :75-107: the same statements as orc_count_interval */
	if (!orc_count_interval(flag, mapq, tid, mtid, pos, mpos, tlen, end_pos, p, &start, &end)) return 0;
	while (st->len > 0 && rend[st->deque[st->front]] < (uint32_t)st->prev_pos) { st->front++; st->len--; }   /* :116-119 */
	orc_count_overlaps(rstart, rend, st->deque + st->front, st->len, start, end, region_frags);            /* :122-126 */
	return 0;
}

int orc_count_batch(orc_count_state *st, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid,
                    const int32_t *mtid, const int32_t *pos, const int32_t *mpos, const int32_t *tlen,
                    const int32_t *end_pos, int64_t n, const orc_count_params *p, int32_t n_chr,
                    const int32_t *rchr, const uint32_t *rstart, const uint32_t *rend, int64_t n_regions,
                    uint32_t *region_frags, int64_t *where)
{
	for (int64_t i = 0; i < n; i++) {
		int rc = orc_count_record(st, flag[i], mapq[i], tid[i], mtid[i], pos[i], mpos[i], tlen[i], end_pos ? end_pos[i] : pos[i], p,
		                          n_chr, rchr, rstart, rend, n_regions, region_frags);
		if (rc) { if (where) *where = i; return rc; }
	}
	return 0;
}

/* ---- fasta gc content: src/fasta_gc_content.rs:44-45 ---------------------------------- */
void orc_gc_count(const uint8_t *seq, size_t n, uint64_t *gc, uint64_t *total)
{
	uint64_t g = 0, t = 0;
	for (size_t i = 0; i < n; i++) {
		uint8_t b = seq[i];
		if (b == 'C' || b == 'G' || b == 'c' || b == 'g') g++;
		if (b != 'N' && b != 'n') t++;
	}
	*gc = g;
	*total = t;
}
