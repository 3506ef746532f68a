// Reads a file through host::BgzfStream and writes what it got to stdout; the exit status says how the stream ended
// (0 = end of data, 3 = corrupt).  Driven by tests/test_cli_cpu.py and tests/test_bam_spec.py.
//   bgzf_stream_test <file>                      the bytes, read in odd-sized pieces
//   bgzf_stream_test <file> records <skip> [slow] BAM records after <skip> header bytes, one line of core fields each:
//                                                by bam_records() with read()/skip() for the records it leaves (as the `sam`
//                                                host does), or — slow — by read()/skip() alone; then how the stream ended
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_common.h"

static uint32_t le32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static int records(host::BgzfStream &s, size_t skip, bool slow)
{
	std::vector<unsigned char> buf(1 << 16);
	while (skip) {
		const long r = s.read(buf.data(), skip < buf.size() ? skip : buf.size());
		if (r <= 0) return 3;
		skip -= (size_t)r;
	}
	std::vector<host::BgzfStream::BamRec> recs;
	size_t n = 0, fast = 0;
	for (;;) {
		recs.clear();
		if (!slow && s.bam_records(recs) > 0) {
			for (const auto &r : recs) printf("%d %d %u %d %d %d %u %u\n", r.tid, r.pos, r.flag, r.mtid, r.mpos, r.tlen, r.mapq, r.l_read_name);
			n += recs.size(); fast += recs.size();
			continue;
		}
		unsigned char hc[36];
		long got = 0;
		while (got < 36) {                                    // one record the long way (the `sam` host's BamStream::next)
			const long r = s.read(hc + got, (size_t)(36 - got));
			if (r < 0) { printf("end: invalid after %zu records\n", n); return 3; }
			if (r == 0) break;
			got += r;
		}
		if (got == 0) { printf("end: clean after %zu records\n", n); break; }
		if (got < 4) { printf("end: premature after %zu records\n", n); return 4; }
		const uint32_t block_size = le32(hc);
		if (block_size < 32) { printf("end: invalid record after %zu records\n", n); return 5; }
		if (got < 36) { printf("end: premature after %zu records\n", n); return 4; }
		size_t rest = block_size - 32;
		while (rest) {
			const long r = s.skip(rest);
			if (r < 0) { printf("end: invalid after %zu records\n", n); return 3; }
			if (r == 0) { printf("end: premature after %zu records\n", n); return 4; }
			rest -= (size_t)r;
		}
		const unsigned char *c = hc + 4;                      // a record counts once all of it was there
		printf("%d %d %u %d %d %d %u %u\n", (int32_t)le32(c), (int32_t)le32(c + 4), (unsigned)(c[14] | (c[15] << 8)), (int32_t)le32(c + 20), (int32_t)le32(c + 24),
		       (int32_t)le32(c + 28), (unsigned)c[9], (unsigned)c[8]);
		n++;
	}
	fprintf(stderr, "%zu records, %zu through bam_records\n", n, fast);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	const int fd = open(argv[1], O_RDONLY);
	if (fd < 0) return 2;
	if (argc >= 4 && !strcmp(argv[2], "records")) {
		host::BgzfStream s(fd, true);
		const int rc = records(s, (size_t)atol(argv[3]), argc >= 5 && !strcmp(argv[4], "slow"));
		fflush(stdout);
		return rc;
	}
	host::BgzfStream s(fd);
	std::vector<unsigned char> buf(1 << 20);
	size_t sizes[] = {1, 4, 32, 17, 65536, 100000, 3, 999983};
	size_t k = 0;
	for (;;) {
		const size_t want = sizes[k++ % 8];
		const long r = s.read(buf.data(), want);
		if (r < 0) return 3;
		if (r > 0) fwrite(buf.data(), 1, (size_t)r, stdout);
		if ((size_t)r < want) {
			if (s.read(buf.data(), 1) < 0) return 3;       // a short read is either the end of the data or the bytes before an error
			break;
		}
	}
	fflush(stdout);
	return 0;
}
