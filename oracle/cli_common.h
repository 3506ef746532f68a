/*
 * cli_common.h — plumbing for the ORACLE command-line restatements (test infrastructure,
 * parity unpinned; see seqkit_oracle.h).  Restates src/common.rs:11-22,49-112 of the
 * reference: error!, parse_args (docopt grammar restated by hand), FileReader, GzipWriter.
 */
#ifndef ORACLE_CLI_COMMON_H
#define ORACLE_CLI_COMMON_H

#include <errno.h>
#include <fcntl.h>
#include <spawn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "seqkit_oracle.h"

/* ---- error! (src/common.rs:11-16): "ERROR: " + message + newline on stderr, exit(-1) */
static void oc_wait_children(void);
static void oc_error(const char *fmt, ...)
{
	va_list ap;
	fflush(stdout);
	fputs("ERROR: ", stderr);
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fputc('\n', stderr);
	oc_wait_children();
	exit(255);
}

/* a Rust panic (unwrap on Err, failed assert!, slice out of range): status 101 */
static void oc_panic(const char *what)
{
	fflush(stdout);
	fprintf(stderr, "thread 'main' panicked: %s\n", what);
	oc_wait_children();
	exit(101);
}

/* ---- growable byte string --------------------------------------------------------- */
typedef struct { uint8_t *p; size_t n, cap; } oc_str;

static void oc_reserve(oc_str *s, size_t need)
{
	if (need <= s->cap) return;
	size_t c = s->cap ? s->cap : 256;
	while (c < need) c *= 2;
	s->p = (uint8_t *)realloc(s->p, c);
	if (!s->p) { fputs("oracle: out of memory\n", stderr); exit(2); }
	s->cap = c;
}
static void oc_clear(oc_str *s) { s->n = 0; }
static void oc_append(oc_str *s, const void *b, size_t n)
{
	oc_reserve(s, s->n + n + 1);
	memcpy(s->p + s->n, b, n);
	s->n += n;
	s->p[s->n] = 0;
}
static void oc_assign(oc_str *s, const void *b, size_t n) { oc_clear(s); oc_append(s, b, n); }
static int oc_starts_with(const oc_str *s, char c) { return s->n > 0 && s->p[0] == (uint8_t)c; }
/* ---- HashMap<String, u64> with `*map.entry(key).or_insert(0) += 1` ------------------------
 * entries are kept in first-seen order (Rust's HashMap iterates in an arbitrary order; see
 * oracle/seqkit_oracle.h, orc_census).                                                    */
typedef struct { oc_str key; uint64_t count; } oc_count;
typedef struct { oc_count *ent; size_t n, cap; int64_t *slot; size_t nslot; } oc_countmap;

static uint64_t oc_hash(const uint8_t *p, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
	return h;
}
static void oc_countmap_rehash(oc_countmap *m, size_t nslot)
{
	free(m->slot);
	m->slot = (int64_t *)malloc(nslot * sizeof *m->slot);
	if (!m->slot) { fputs("oracle: out of memory\n", stderr); exit(2); }
	m->nslot = nslot;
	for (size_t i = 0; i < nslot; i++) m->slot[i] = -1;
	for (size_t e = 0; e < m->n; e++) {
		size_t i = oc_hash(m->ent[e].key.p, m->ent[e].key.n) & (nslot - 1);
		while (m->slot[i] >= 0) i = (i + 1) & (nslot - 1);
		m->slot[i] = (int64_t)e;
	}
}
static void oc_countmap_add(oc_countmap *m, const uint8_t *key, size_t n)
{
	if (m->nslot == 0) oc_countmap_rehash(m, 1024);
	size_t i = oc_hash(key, n) & (m->nslot - 1);
	while (m->slot[i] >= 0) {
		oc_count *e = &m->ent[m->slot[i]];
		if (e->key.n == n && memcmp(e->key.p, key, n) == 0) { e->count += 1; return; }
		i = (i + 1) & (m->nslot - 1);
	}
	if (m->n == m->cap) {
		m->cap = m->cap ? m->cap * 2 : 256;
		m->ent = (oc_count *)realloc(m->ent, m->cap * sizeof *m->ent);
		if (!m->ent) { fputs("oracle: out of memory\n", stderr); exit(2); }
	}
	memset(&m->ent[m->n], 0, sizeof(oc_count));
	oc_assign(&m->ent[m->n].key, key, n);
	m->ent[m->n].count = 1;
	m->slot[i] = (int64_t)m->n++;
	if (m->n * 2 > m->nslot) oc_countmap_rehash(m, m->nslot * 2);
}

/* String::drain(start..end) */
static void oc_drain(oc_str *s, size_t start, size_t end)
{
	memmove(s->p + start, s->p + end, s->n - end);
	s->n -= (end - start);
	if (s->p) s->p[s->n] = 0;
}

/* ---- child process bookkeeping ---------------------------------------------------- */
static pid_t oc_children[1024];
static int oc_nchildren = 0;
static FILE *oc_child_pipes[1024];
static int oc_nchild_pipes = 0;

/* The reference never wait()s for its gzip children (they finish on their own after the
 * parent exits).  The oracle closes the pipes and waits so that a test can read complete
 * files right after the process returns; the bytes are the same.                         */
static void oc_wait_children(void)
{
	for (int i = 0; i < oc_nchild_pipes; i++) if (oc_child_pipes[i]) fclose(oc_child_pipes[i]);
	oc_nchild_pipes = 0;
	for (int i = 0; i < oc_nchildren; i++) { int st; waitpid(oc_children[i], &st, 0); }
	oc_nchildren = 0;
}

/* ---- FileReader (src/common.rs:83-112) ------------------------------------------- */
typedef struct { FILE *f; } oc_reader;

static oc_reader oc_reader_open(const char *path)
{
	oc_reader r;
	if (strcmp(path, "-") == 0) { r.f = stdin; return r; }
	int fd = open(path, O_RDONLY);
	if (fd < 0) oc_error("Cannot open file %s for reading.", path);
	size_t pl = strlen(path);
	if (pl >= 3 && strcmp(path + pl - 3, ".gz") == 0) {
		int pp[2];
		if (pipe(pp) != 0) oc_error("Cannot start gunzip process.");
		pid_t pid = fork();
		if (pid < 0) oc_error("Cannot start gunzip process.");
		if (pid == 0) {
			dup2(fd, 0); dup2(pp[1], 1);
			close(fd); close(pp[0]); close(pp[1]);
			execlp("gunzip", "gunzip", "-c", (char *)NULL);
			_exit(127);
		}
		close(fd); close(pp[1]);
		if (oc_nchildren < 1024) oc_children[oc_nchildren++] = pid;
		r.f = fdopen(pp[0], "rb");
	} else {
		r.f = fdopen(fd, "rb");
	}
	if (!r.f) oc_error("Cannot open file %s for reading.", path);
	setvbuf(r.f, NULL, _IOFBF, 1 << 16);
	return r;
}

/* read_line: clear, read through '\n' (kept), false at EOF; invalid UTF-8 is an error. */
static int oc_read_line(oc_reader *r, oc_str *line)
{
	oc_clear(line);
	oc_reserve(line, 1);
	line->p[0] = 0;
	int c;
	while ((c = getc_unlocked(r->f)) != EOF) {
		if (line->n + 2 > line->cap) oc_reserve(line, line->n + 2);
		line->p[line->n++] = (uint8_t)c;
		if (c == '\n') break;
	}
	if (ferror(r->f)) oc_error("I/O error while reading from file.");
	line->p[line->n] = 0;
	if (!orc_utf8_valid(line->p, line->n)) oc_error("I/O error while reading from file.");
	return line->n > 0;
}

/* ---- GzipWriter (src/common.rs:49-81): File::create + `gzip -c` / `pigz -c` child -- */
static FILE *oc_gzip_writer(const char *path, int use_pigz)
{
	int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
	if (fd < 0) oc_error("Cannot open file %s for writing.", path);
	int pp[2];
	const char *prog = use_pigz ? "pigz" : "gzip";
	if (pipe2(pp, O_CLOEXEC) != 0) oc_error("Cannot start %s process.", prog);
	/* Command::spawn() fails in the parent when the program does not exist: posix_spawnp reports that too */
	posix_spawn_file_actions_t fa;
	posix_spawn_file_actions_init(&fa);
	posix_spawn_file_actions_adddup2(&fa, pp[0], 0);
	posix_spawn_file_actions_adddup2(&fa, fd, 1);
	char *const av[] = {(char *)prog, (char *)"-c", NULL};
	pid_t pid;
	extern char **environ;
	int rc = posix_spawnp(&pid, prog, &fa, NULL, av, environ);
	posix_spawn_file_actions_destroy(&fa);
	if (rc != 0) oc_error("Cannot start %s process.", prog);
	close(fd); close(pp[0]);
	if (oc_nchildren < 1024) oc_children[oc_nchildren++] = pid;
	FILE *f = fdopen(pp[1], "wb");
	if (!f) oc_error("Cannot start %s process.", prog);
	if (oc_nchild_pipes < 1024) oc_child_pipes[oc_nchild_pipes++] = f;
	return f;
}

/* ---- docopt grammar, restated (src/common.rs:18-22) ------------------------------- */
typedef struct {
	const char *name;      /* "--dry-run" */
	int takes_value;
	const char *value;     /* NULL = absent; for flags "" = present */
} oc_opt;

/* Parses argv[first..) into options (long options only; unique-prefix matching, `--o=v`
 * and `--o v`, `--` ends options, `-` is a positional) and positionals.  Returns 0 on a
 * grammar violation (caller prints "Invalid arguments.\n<usage>").                        */
static int oc_parse(int argc, char **argv, int first, oc_opt *opts, int nopts,
                    const char **pos, int *npos, int maxpos)
{
	int only_pos = 0;
	*npos = 0;
	for (int i = first; i < argc; i++) {
		const char *a = argv[i];
		if (!only_pos && strcmp(a, "--") == 0) { only_pos = 1; continue; }
		if (!only_pos && a[0] == '-' && a[1] == '-') {
			const char *eq = strchr(a, '=');
			size_t nl = eq ? (size_t)(eq - a) : strlen(a);
			int hit = -1, nh = 0;
			for (int k = 0; k < nopts; k++) {
				if (strlen(opts[k].name) == nl && strncmp(opts[k].name, a, nl) == 0) { hit = k; nh = 1; break; }
				if (strncmp(opts[k].name, a, nl) == 0) { hit = k; nh++; }
			}
			if (nh != 1) return 0;
			if (opts[hit].takes_value) {
				if (eq) opts[hit].value = eq + 1;
				else if (i + 1 < argc) opts[hit].value = argv[++i];
				else return 0;
			} else {
				if (eq) return 0;
				opts[hit].value = "";
			}
			continue;
		}
		if (!only_pos && a[0] == '-' && a[1] != 0) return 0;   /* no short options exist */
		if (*npos >= maxpos) return 0;
		pos[(*npos)++] = a;
	}
	return 1;
}

/* str::parse::<uN>(): optional '+', then >= 1 ASCII digits, no overflow. */
static int oc_parse_uint(const char *s, uint64_t max, uint64_t *out)
{
	if (*s == '+') s++;
	if (!*s) return 0;
	uint64_t v = 0;
	for (; *s; s++) {
		if (*s < '0' || *s > '9') return 0;
		uint64_t d = (uint64_t)(*s - '0');
		if (v > (max - d) / 10) return 0;
		v = v * 10 + d;
	}
	*out = v;
	return 1;
}

/* "{:.1}" of an f64 as Rust prints it (NaN / inf spellings differ from C). */
static void oc_fmt_pct(char *buf, size_t n, double v)
{
	if (v != v) snprintf(buf, n, "NaN");
	else if (v > 1.7e308) snprintf(buf, n, "inf");
	else if (v < -1.7e308) snprintf(buf, n, "-inf");
	else snprintf(buf, n, "%.1f", v);
}

#endif
