// Reads a file through host::BgzfStream in odd-sized pieces and writes what it got to stdout; the exit status says how
// the stream ended (0 = end of data, 3 = corrupt).  Driven by tests/test_cli_cpu.py.
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "host_common.h"

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	const int fd = open(argv[1], O_RDONLY);
	if (fd < 0) return 2;
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
