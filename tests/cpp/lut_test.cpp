// CPU check of the neighbourhood table (seqkit_amd/csrc/sk_lut.{h,cpp}): build tables for many sheets and look every
// observed barcode up with the arithmetic the kernel uses (classify, pack, mix, two probes, separator), against the
// reference's loop written out plainly (src/fasta_demultiplex.rs:154-194, :269-277).  No GPU; the lookup below is a test
// model of demux_lut_kernel, not a product path.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../seqkit_amd/csrc/sk_lut.h"

using sk::LutDev;
using sk::LutHost;

struct Result { int assign, diff, first, last; };


// the factored form (demux_lut_kernel<.., PAIR = true>): each half -> (half id, distance), the pair of ids -> (first, last)
static Result lookup_pair(const LutHost &h, const uint32_t (&c)[5], uint32_t sepbad)
{
	const LutDev &t = h.dev;
	const sk::LutPairDev &pr = t.pair;
	auto probe = [&](uint32_t base, int nb, uint32_t key, uint32_t seed, uint32_t &val) {
		const uint32_t x = sk::lut_mix(key, 0u, seed), m = (1u << nb) - 1u;
		const uint32_t *e1 = &h.slots[2 * (size_t)(base + (x >> (32 - nb)))], *e2 = &h.slots[2 * (size_t)(base + m + 1u + ((x >> (32 - 2 * nb)) & m))];
		// (a half that equals the table's free word meets free slots, maybe on both sides: their value word says distance 65 535)
		if (e1[0] == key && e2[0] == key && !(e1[1] == sk::kLutPairFreeVal && e2[1] == sk::kLutPairFreeVal)) { fprintf(stderr, "key in both tables\n"); exit(1); }
		if (e1[0] == key) { val = e1[1]; return true; }
		if (e2[0] == key) { val = e2[1]; return true; }
		return false;
	};
	const uint32_t A1 = (t.W1 == 1 ? c[0] : sk::lut_pack_half(c[0], c[1], t.wide)) & pr.keep1;
	const uint32_t A2 = (t.W1 == 1 ? c[1] : sk::lut_pack_half(c[2], c[3], t.wide)) & pr.keep2;
	Result r = {-1, 255, -1, -1};
	uint32_t v1 = 0, v2 = 0, pv = 0;
	if (!probe(0u, pr.nb1, A1, pr.seed1, v1) || !probe(pr.off2, pr.nb2, A2, pr.seed2, v2)) return r;
	const int tot = (int)(v1 >> 16) + (int)(v2 >> 16) + (int)sepbad;
	if (tot > t.max_diff) return r;
	if (!probe(pr.offp, pr.nbp, (v1 & 0x3ffu) | ((v2 & 0x3ffu) << 10), pr.seedp, pv)) return r;
	r.diff = tot; r.first = (int)(pv & 0xffffu); r.last = (int)(pv >> 16);
	r.assign = r.first == r.last ? r.first : -2;
	return r;
}

static Result lookup(const LutHost &h, const uint8_t *obs, int L)
{
	const LutDev &t = h.dev;
	uint32_t d[5] = {0, 0, 0, 0, 0}, c[5] = {0, 0, 0, 0, 0};
	uint8_t padded[32];
	memset(padded, 0xEE, sizeof padded);                                   // bytes past L are garbage to the kernel too
	memcpy(padded, obs, (size_t)L);
	for (int w = 0; w < t.W1; w++) memcpy(&d[w], padded + 4 * w, 4);
	for (int w = 0; w < t.W2; w++) memcpy(&d[t.W1 + w], padded + t.sep_off + 1 + 4 * w, 4);
	const uint32_t sepbad = (t.sep_off >= 0 && padded[t.sep_off] != t.sep_val) ? 1u : 0u;
	for (int w = 0; w < t.W1 + t.W2; w++) c[w] = sk::lut_classes(d[w], t);      // (the function the kernels call; its byte permutes are loops here)
	if (t.pair.bytes != 0) return lookup_pair(h, c, sepbad);
	uint32_t A, B;
	sk::lut_pack(c, A, B, t.wide);
	A &= t.keepA; B &= t.keepB;
	const uint32_t x = sk::lut_mix(A, B, t.seed);
	const uint32_t y = (x << t.nb) | (x >> (32 - t.nb));          // table 2: the nb bits below table 1's
	const uint32_t *e1 = &h.slots[2 * (size_t)(x >> (32 - t.nb))];
	const uint32_t *e2 = &h.slots[2 * ((size_t)t.mask + 1 + (y >> (32 - t.nb)))];
	auto hit = [&](const uint32_t *e, uint32_t tag) { return (((e[0] ^ B) & 0x7fffffffu) | ((e[1] ^ tag) & t.tag_mask)) == 0; };
	const bool h1 = hit(e1, x), h2 = hit(e2, y);
	if (h1 && h2) { fprintf(stderr, "key in both tables\n"); exit(1); }
	Result r = {-1, 255, -1, -1};
	if (!h1 && !h2) return r;
	const uint32_t *e = h1 ? e1 : e2;
	const int tot = (int)(e[0] >> 31) + (int)sepbad;
	if (tot > t.max_diff) return r;
	const int idx = (int)((e[1] >> t.idx_shift) & t.idx_mask);
	r.diff = tot;
	if (e[1] >> 31) { r.assign = -2; r.first = h.amb[2 * (size_t)idx]; r.last = h.amb[2 * (size_t)idx + 1]; }
	else { r.assign = idx; r.first = r.last = idx; }
	return r;
}

static Result reference(const std::vector<uint8_t> &sheet, int S, int L, int max_diff, const uint8_t *obs)
{
	int lowest = 0x7fffffff, first = 0, last = 0;
	for (int s = 0; s < S; s++) {
		int d = 0;
		for (int k = 0; k < L; k++) {
			const uint8_t cb = sheet[(size_t)s * L + k];
			if (cb == 'N' || cb == 'U') continue;
			d += obs[k] != cb;
		}
		if (d < lowest) { lowest = d; first = s; last = s; }
		else if (d == lowest) last = s;
	}
	Result r = {-1, lowest, first, last};
	if (lowest <= max_diff) r.assign = first == last ? first : -2;
	return r;
}

int main(int argc, char **argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 300;
	std::mt19937_64 rng(12345);
	auto pick = [&](int n) { return (int)(rng() % (uint64_t)n); };
	const char *alphabets[] = {"ACGT", "ACGTN", "ACGT+", "ACGTacg", "AC", "ACGTN+U", "ACGT-_", "ACGTRYK", "ACGTacgt", "ACGTNacgtn", "ACGTacgtRY", "ACGTRYKMSWBDHV"};
	const int n_alpha = 12;
	int built = 0, refused = 0, factored = 0, wide = 0, wide_classes = 0;
	size_t checked = 0;
	// half-barcodes of `hl` letters, pairwise at least `dist` apart (greedy)
	auto distant = [&](int count, int hl, int dist) {
		std::vector<std::string> out;
		for (int tries = 0; (int)out.size() < count && tries < 200000; tries++) {
			std::string c((size_t)hl, 'A');
			for (auto &ch : c) ch = "ACGT"[pick(4)];
			bool ok = true;
			for (const auto &o : out) { int d = 0; for (int k = 0; k < hl; k++) d += c[(size_t)k] != o[(size_t)k]; if (d < dist) { ok = false; break; } }
			if (ok) out.push_back(c);
		}
		return out;
	};
	for (int it = 0; it < rounds; it++) {
		const int Ss[] = {1, 2, 3, 16, 40, 96, 128, 129, 384, 1000};
		const int Ls[] = {1, 3, 4, 8, 9, 12, 17, 20};
		int S = Ss[pick(it % 8 == 0 ? 10 : 7)], L = Ls[pick(8)];
		const int ai = it % 5 == 2 ? 8 + pick(n_alpha - 8) : pick(8);          // every fifth sheet from an alphabet of more than seven letters
		const std::string al = alphabets[ai];
		if (ai >= 8) L = Ls[pick(4)];                                          // (wide classes serve at most 8 columns per word: short rows, or the dual sheets below)
		std::vector<uint8_t> sheet((size_t)S * L);
		for (auto &b : sheet) b = (uint8_t)al[(size_t)pick((int)al.size())];
		int kind = pick(5);
		if (it % 4 == 1) {
			// a dual-index sheet `i7+i5` out of two sets of half-barcodes: combinations of n7 x n5, some twice, halves >= 3 apart
			// (the factored form's case when the full-key table is large) or only >= 1 apart (then it must fall back)
			const int hl = pick(2) ? 8 : 4, dist = pick(4) ? 3 : 1;
			const int want = hl == 8 ? (pick(2) ? 384 : (pick(2) ? 1000 : 200)) : 40;
			const int n7 = 2 + pick(hl == 8 ? 60 : 6), n5 = 1 + (want + n7 - 1) / n7;
			const auto i7 = distant(n7, hl, dist), i5 = distant(n5, hl, dist);
			S = std::min(want, (int)(i7.size() * i5.size()));
			L = 2 * hl + 1;
			sheet.assign((size_t)S * L, 0);
			for (int s2 = 0; s2 < S; s2++) {
				const int combo = pick(7) == 0 ? pick((int)(i7.size() * i5.size())) : s2;      // mostly distinct combinations, some repeated
				memcpy(&sheet[(size_t)s2 * L], i7[(size_t)combo % i7.size()].data(), (size_t)hl);
				sheet[(size_t)s2 * L + hl] = '+';
				memcpy(&sheet[(size_t)s2 * L + hl + 1], i5[(size_t)combo / i7.size()].data(), (size_t)hl);
			}
			kind = 0;
			if (pick(4) == 0) for (int s2 = 0; s2 < S; s2++) sheet[(size_t)s2 * L + L - 1] = 'U';      // a UMI column at the end
			// every other row typed in lower case: eight letters — wide classes, served by the factored form or not at all
			if (it % 8 == 5) for (int s2 = 1; s2 < S; s2 += 2) for (int k = 0; k < L; k++) { uint8_t &b = sheet[(size_t)s2 * L + k]; if (b >= 'A' && b <= 'T' && b != 'U') b = (uint8_t)(b + 32); }
		}
		if (kind == 1 && L >= 4) for (int s = 0; s < S; s++) for (int k = L - 3; k < L; k++) sheet[(size_t)s * L + k] = 'U';
		if (kind == 2) for (auto &b : sheet) if (pick(10) == 0) b = 'N';
		// wildcards of single rows beside a separator: sprinkled BEFORE the separator column is written (kind 4) and into the letter
		// columns of a dual-index sheet — the per-class enumeration of such rows together with sep_off / the keep masks, and the
		// `mixed` gate that keeps such a sheet out of the factored form
		const bool sprinkle = (kind == 4 || (it % 4 == 1 && it % 3 == 0)) && pick(2) == 0;
		if (sprinkle) for (auto &b : sheet) if (b != '+' && pick(12) == 0) b = pick(2) ? 'N' : 'U';
		if (kind >= 3 && L >= 3) { const int c = pick(L); for (int s = 0; s < S; s++) sheet[(size_t)s * L + c] = '+'; }      // a separator (when '+' is nowhere else)
		if (S >= 3 && pick(3) == 0) memcpy(&sheet[2 * (size_t)L], &sheet[0], (size_t)L);                                       // duplicate rows
		const int max_diff = pick(4) == 0 ? 0 : 1;
		LutHost h;
		if (!sk::lut_build(sheet.data(), S, L, max_diff, h)) { refused++; continue; }
		built++;
		factored += h.dev.pair.bytes != 0;
		wide_classes += h.dev.wide != 0;
		wide += h.dev.pair.bytes == 0 && h.dev.idx_shift != 24;
		const char noise[] = "ACGTNacgtn+U\x00\xff#-_";
		for (int r = 0; r < 3000; r++) {
			uint8_t obs[32];
			if (pick(10) == 0) for (int k = 0; k < L; k++) obs[k] = (uint8_t)noise[pick((int)sizeof noise - 1)];
			else {
				memcpy(obs, &sheet[(size_t)pick(S) * L], (size_t)L);
				const int nsub = pick(4);
				for (int j = 0; j < nsub; j++) obs[pick(L)] = (uint8_t)noise[pick((int)sizeof noise - 1)];
			}
			// deterministic probes first: ONE byte repeated over a whole half (or the whole row) beside a valid other half — with
			// 4-bit classes eight bytes of class 15 pack to 0xFFFFFFFF, the word a free slot of the factored form used to hold alone
			// (a poly-G index read on a sheet typed in both cases was given the sample of half 0)
			const std::string probes = std::string(noise, sizeof noise - 1) + al;      // every noise byte and every letter of the sheet's alphabet
			const int n_noise = (int)probes.size();
			if (r < 3 * n_noise) {
				const int hl = h.dev.sep_off >= 0 ? h.dev.sep_off : L;
				memcpy(obs, &sheet[(size_t)pick(S) * L], (size_t)L);
				const int part = r / n_noise, lo = part == 1 ? L - hl : 0, hi = part == 0 ? hl : L;
				for (int k = lo; k < hi; k++) if (k != h.dev.sep_off) obs[k] = (uint8_t)probes[(size_t)(r % n_noise)];
			}
			const Result got = lookup(h, obs, L), want = reference(sheet, S, L, max_diff, obs);
			checked++;
			const bool same = got.assign == want.assign && (want.assign == -1 || (got.diff == want.diff && got.first == want.first && got.last == want.last));
			if (!same) {
				fprintf(stderr, "MISMATCH it=%d S=%d L=%d max_diff=%d: got (%d,%d,%d,%d) want (%d,%d,%d,%d)\n", it, S, L, max_diff,
				        got.assign, got.diff, got.first, got.last, want.assign, want.diff, want.first, want.last);
				return 1;
			}
		}
	}
	{	// rows with a wildcard where other rows have a letter (SURVEY.md Appendix A's pair among them) are served by the table
		const char *rows[] = {"ACGTACGT", "ACGTNCGT", "TTGCAANN", "NGGATCCA", "CATGCATG"};
		std::vector<uint8_t> sheet;
		for (const char *r : rows) sheet.insert(sheet.end(), r, r + 8);
		LutHost h;
		if (!sk::lut_build(sheet.data(), 5, 8, 1, h)) { fprintf(stderr, "a sheet with wildcards in some rows was refused\n"); return 1; }
		const char *obs[] = {"ACGTACGT", "ACGTTCGT", "ACGTNCGT", "TTGCAAAC", "TTGCATGG", "AGGATCCA", "NGGATCCN", "CATGCATG", "CATGCANN"};
		for (const char *o : obs) {
			const Result got = lookup(h, (const uint8_t *)o, 8), want = reference(sheet, 5, 8, 1, (const uint8_t *)o);
			if (got.assign != want.assign || (want.assign != -1 && (got.diff != want.diff || got.first != want.first || got.last != want.last))) {
				fprintf(stderr, "MISMATCH on %s: got (%d,%d,%d,%d) want (%d,%d,%d,%d)\n", o, got.assign, got.diff, got.first, got.last, want.assign, want.diff, want.first, want.last);
				return 1;
			}
		}
	}
	// the two sheets of the benchmark: sizes (informative)
	printf("ok: %d tables built (%d factored, %d with 10-bit sample indices, %d with wide classes), %d sheets refused, %zu lookups\n", built, factored, wide, wide_classes, refused, checked);
	return 0;
}
