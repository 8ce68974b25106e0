/* seqio.c -- FASTA and PHYLIP alignment readers (the on-disk
 * formats in front of the hot path).
 *
 * Replaces fasta.c (pll_fasta_open :40, _rewind :102, _close :124, _getnext :130,
 * _getfilesize :316, _getfilepos :321) and phylip.c (pll_phylip_open :282,
 * _rewind :345, _close :368, _parse_interleaved :376, _parse_sequential :568,
 * pll_msa_destroy :705) of the reference.  Host-only text handling; written from
 * the behaviour of those files, and checked against them (built into
 * oracle/_ref/libpll_ref.so) on a corpus of well-formed and malformed inputs in
 * tests/test_seqio.py: same records, same counters, same pll_errno / pll_errmsg.
 *
 * Behaviour that callers can observe and that is therefore kept:
 *  - every input character is classified by the caller's 256-entry table
 *    (pll_map_fasta / pll_map_phylip): 0 = dropped and counted in stripped[],
 *    1 = data, 2 = fatal, 3 = dropped silently;
 *  - FASTA text is consumed in PLL_LINEALLOC-sized pieces, so a header longer
 *    than one piece is cut there and its tail is read as sequence data;
 *  - the FASTA record buffers belong to the caller after a successful call, and
 *    are left allocated when a fatal character stops the parse;
 *  - PHYLIP keeps lineno at 1 and reports illegal characters as being "in the
 *    fasta file" -- the reference's wording is part of the message contract;
 *  - a PHYLIP header with anything after the two numbers fails without touching
 *    pll_errno.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "internal.h"

#define RECORD_CHUNK 4096

static void out_of_memory(void)
{
  pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
}

/* text up to (not including) the first `stop`, or all of it */
static long span_until(const char * s, int stop)
{
  const char * hit = strchr(s, stop);
  return hit ? (long)(hit - s) : (long)strlen(s);
}

static void report_fatal_char(int fasta, char c, long lineno)
{
  if ((unsigned char)c >= 32)
    pll_amd_set_error(fasta ? PLL_ERROR_FASTA_ILLEGALCHAR : PLL_ERROR_PHYLIP_ILLEGALCHAR,
                      "illegal character '%c' on line %ld in the fasta file", c, lineno);
  else
    pll_amd_set_error(fasta ? PLL_ERROR_FASTA_UNPRINTABLECHAR : PLL_ERROR_PHYLIP_UNPRINTABLECHAR,
                      "illegal unprintable character %#.2x (hexadecimal) on line %ld in the fasta file",
                      (unsigned int)(unsigned char)c, lineno);
}

static void clear_stripped(long * count, long * per_char)
{
  *count = 0;
  memset(per_char, 0, 256 * sizeof(long));
}

/* file size by seeking to the end; leaves the stream at the start */
static int measure_file(FILE * fp, long * size)
{
  if (fseek(fp, 0, SEEK_END)) return 0;
  *size = ftell(fp);
  rewind(fp);
  return 1;
}

/* ------------------------------------------------------------------ FASTA */

/* next piece of text into fd->line ("" at end of file) */
static void fasta_fill(pll_fasta_t * fd)
{
  fd->line[0] = 0;
  if (!fgets(fd->line, PLL_LINEALLOC, fd->fp)) fd->line[0] = 0;
}

pll_fasta_t * pll_fasta_open(const char * filename, const unsigned int * map)
{
  pll_fasta_t * fd = (pll_fasta_t *)malloc(sizeof(pll_fasta_t));
  if (!fd)
  {
    out_of_memory();
    return NULL;
  }
  fd->lineno = 0;
  fd->no = -1;
  fd->chrstatus = map;
  fd->fp = fopen(filename, "r");
  if (!fd->fp)
  {
    pll_amd_set_error(PLL_ERROR_FILE_OPEN, "Unable to open file (%s)", filename);
    free(fd);
    return NULL;
  }
  if (!measure_file(fd->fp, &fd->filesize))
  {
    pll_amd_set_error(PLL_ERROR_FILE_SEEK, "Unable to seek in file (%s)", filename);
    fclose(fd->fp);
    free(fd);
    return NULL;
  }
  clear_stripped(&fd->stripped_count, fd->stripped);
  fasta_fill(fd);
  if (!fd->line[0])
  {
    pll_amd_set_error(PLL_ERROR_FILE_SEEK, "Unable to read file (%s)", filename);
    fclose(fd->fp);
    free(fd);
    return NULL;
  }
  fd->lineno = 1;
  return fd;
}

int pll_fasta_rewind(pll_fasta_t * fd)
{
  rewind(fd->fp);
  clear_stripped(&fd->stripped_count, fd->stripped);
  fasta_fill(fd);
  if (!fd->line[0])
  {
    pll_amd_set_error(PLL_ERROR_FILE_SEEK, "Unable to rewind and cache data");
    return PLL_FAILURE;
  }
  fd->lineno = 1;
  return PLL_SUCCESS;
}

void pll_fasta_close(pll_fasta_t * fd)
{
  fclose(fd->fp);
  free(fd);
}

long pll_fasta_getfilesize(const pll_fasta_t * fd) { return fd->filesize; }

long pll_fasta_getfilepos(pll_fasta_t * fd) { return ftell(fd->fp); }

/* a growing, caller-owned byte buffer */
typedef struct
{
  char * data;
  long used, room;
} record_buf;

static int record_reserve(record_buf * b, long need, long step)
{
  if (need <= b->room) return 1;
  long room = step ? b->room + step : need;
  while (room < need) room += step ? step : 1;
  char * grown = (char *)realloc(b->data, (size_t)room);
  if (!grown) return 0;
  b->data = grown;
  b->room = room;
  return 1;
}

int pll_fasta_getnext(pll_fasta_t * fd, char ** head, long * head_len, char ** seq, long * seq_len,
                      long * seqno)
{
  record_buf h = {(char *)malloc(RECORD_CHUNK), 0, RECORD_CHUNK};
  record_buf s = {NULL, 0, RECORD_CHUNK};
  *head_len = 0;
  *seq_len = 0;
  *head = h.data;
  if (!h.data)
  {
    out_of_memory();
    return PLL_FAILURE;
  }
  s.data = (char *)malloc(RECORD_CHUNK);
  *seq = s.data;
  if (!s.data)
  {
    free(h.data);
    out_of_memory();
    return PLL_FAILURE;
  }

  if (!fd->line[0])
  {
    pll_amd_set_error(PLL_ERROR_FILE_EOF, "End of file\n");
    free(h.data);
    free(s.data);
    return PLL_FAILURE;
  }
  if (fd->line[0] != '>')
  {
    pll_amd_set_error(PLL_ERROR_FASTA_INVALIDHEADER, "Illegal header line in query fasta file");
    free(h.data);
    free(s.data);
    return PLL_FAILURE;
  }

  /* header: the rest of this piece up to a CR if there is one, else up to the LF */
  const char * text = fd->line + 1;
  h.used = span_until(text, strchr(text, '\r') ? '\r' : '\n');
  if (!record_reserve(&h, h.used + 1, 0))
  {
    out_of_memory();
    free(h.data);
    free(s.data);
    return PLL_FAILURE;
  }
  memcpy(h.data, text, (size_t)h.used);
  h.data[h.used] = 0;
  *head = h.data;
  *head_len = h.used;

  /* sequence: every following piece that does not start a new record */
  for (fasta_fill(fd), fd->lineno++; fd->line[0] && fd->line[0] != '>'; fasta_fill(fd), fd->lineno++)
  {
    for (const char * p = fd->line; *p; ++p)
    {
      /* bytes >= 0x80 index the table's upper half (the reference's `(int)c` of a signed
         char is a negative index there: fasta.c:219) */
      const char c = *p;
      switch ((char)fd->chrstatus[(unsigned char)c])
      {
        case 0:
          fd->stripped_count++;
          fd->stripped[(unsigned char)c]++;
          break;
        case 1:
          if (!record_reserve(&s, s.used + 1, RECORD_CHUNK))
          {
            out_of_memory();
            free(h.data);
            free(s.data);
            return PLL_FAILURE;
          }
          s.data[s.used++] = c;
          *seq = s.data;
          *seq_len = s.used;
          break;
        case 2:
          report_fatal_char(1, c, fd->lineno);
          return PLL_FAILURE; /* both buffers stay with the caller, as in the reference */
        default:
          break;
      }
    }
  }

  if (!record_reserve(&s, s.used + 1, RECORD_CHUNK))
  {
    out_of_memory();
    free(h.data);
    free(s.data);
    return PLL_FAILURE;
  }
  s.data[s.used] = 0;
  *seq = s.data;
  *seq_len = s.used;
  *seqno = ++fd->no;
  return PLL_SUCCESS;
}

/* ----------------------------------------------------------------- PHYLIP */

enum { PHY_SEQUENTIAL, PHY_INTERLEAVED };

static int is_blank(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

/* make room for `room` bytes in fd->line, keeping the line_size bytes already there */
static int phy_line_room(pll_phylip_t * fd, size_t room)
{
  char * grown = (char *)malloc(room);
  if (!grown)
  {
    out_of_memory();
    return 0;
  }
  if (fd->line_size) memcpy(grown, fd->line, fd->line_size);
  free(fd->line);
  fd->line = grown;
  fd->line_maxsize = room;
  return 1;
}

/* One whole text line of any length into fd->line, newline removed; NULL at the
 * end of the file (fd->line is then released) or when memory runs out. */
static char * phy_next_line(pll_phylip_t * fd)
{
  fd->line_size = 0;
  while (fgets(fd->buffer, PLL_LINEALLOC, fd->fp))
  {
    const size_t got = strlen(fd->buffer);
    if (fd->line_size + got > fd->line_maxsize && !phy_line_room(fd, fd->line_maxsize + PLL_LINEALLOC))
      return NULL;
    memcpy(fd->line + fd->line_size, fd->buffer, got);
    fd->line_size += got;
    if (fd->buffer[got - 1] == '\n')
    {
      fd->line[fd->line_size - 1] = 0;
      return fd->line;
    }
  }
  if (!fd->line_size)
  {
    /* (the reference keeps line_maxsize here, so that its pll_phylip_rewind after a
       complete parse copies into a NULL buffer; resetting it makes rewind usable) */
    free(fd->line);
    fd->line = NULL;
    fd->line_maxsize = 0;
    return NULL;
  }
  /* last line without a newline */
  if (fd->line_size == fd->line_maxsize && !phy_line_room(fd, fd->line_maxsize + 1)) return NULL;
  fd->line[fd->line_size] = 0;
  return fd->line;
}

pll_phylip_t * pll_phylip_open(const char * filename, const unsigned int * map)
{
  pll_phylip_t * fd = (pll_phylip_t *)malloc(sizeof(pll_phylip_t));
  if (!fd)
  {
    out_of_memory();
    return NULL;
  }
  fd->line = NULL;
  fd->line_size = 0;
  fd->line_maxsize = 0;
  fd->lineno = 0;
  fd->no = -1;
  fd->chrstatus = map;
  fd->fp = fopen(filename, "r");
  if (!fd->fp)
  {
    pll_amd_set_error(PLL_ERROR_FILE_OPEN, "Unable to open file (%s)", filename);
    free(fd);
    return NULL;
  }
  if (!measure_file(fd->fp, &fd->filesize))
  {
    pll_amd_set_error(PLL_ERROR_FILE_SEEK, "Unable to seek in file (%s)", filename);
    fclose(fd->fp);
    free(fd);
    return NULL;
  }
  clear_stripped(&fd->stripped_count, fd->stripped);
  if (!phy_next_line(fd))
  {
    free(fd->line);
    fclose(fd->fp);
    free(fd);
    return NULL;
  }
  fd->lineno = 1;
  return fd;
}

int pll_phylip_rewind(pll_phylip_t * fd)
{
  rewind(fd->fp);
  clear_stripped(&fd->stripped_count, fd->stripped);
  if (!phy_next_line(fd))
  {
    pll_amd_set_error(PLL_ERROR_FILE_SEEK, "Unable to rewind and cache data");
    return PLL_FAILURE;
  }
  fd->lineno = 1;
  fd->no = -1;
  return PLL_SUCCESS;
}

void pll_phylip_close(pll_phylip_t * fd)
{
  fclose(fd->fp);
  free(fd->line);
  free(fd);
}

void pll_msa_destroy(pll_msa_t * msa)
{
  if (!msa) return;
  for (int pass = 0; pass < 2; ++pass)
  {
    char ** rows = pass ? msa->sequence : msa->label;
    if (!rows) continue;
    for (int i = 0; i < msa->count; ++i) free(rows[i]);
    free(rows);
  }
  free(msa);
}

/* a leading decimal integer; *used = characters consumed (0 if there is none) */
static int leading_int(const char * text, int * used)
{
  int value = 0;
  *used = 0;
  if (sscanf(text, "%d%n", &value, used) < 1 || !*used) return 0;
  return value;
}

/* "<count> <length>" and nothing else.  1 = ok, 0 = refused; pll_errno is set
 * only when one of the two numbers is missing or zero. */
static int phy_header(const char * line, int * count, int * length)
{
  int used;
  if (!(*count = leading_int(line, &used)))
  {
    pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Invalid number of sequences in header");
    return 0;
  }
  line += used;
  if (!(*length = leading_int(line, &used)))
  {
    pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Invalid sequence length in header");
    return 0;
  }
  line += used;
  while (*line && is_blank(*line)) ++line;
  /* the reference accepts an 's'/'i' option letter in interleaved mode only to
     refuse it one statement later: any trailing token fails, in both modes */
  return *line == 0;
}

/* Append the data characters of `text` to row `row` from column `col` on.
 * Returns how many were appended, -1 on error (pll_errno set). */
static int phy_take(pll_phylip_t * fd, pll_msa_t * msa, const char * text, int row, int col)
{
  char * dst = msa->sequence[row] + col;
  int n = 0;
  for (; *text; ++text)
  {
    const char c = *text;
    switch ((char)fd->chrstatus[(unsigned char)c])
    {
      case 0:
        fd->stripped_count++;
        fd->stripped[(unsigned char)c]++;
        break;
      case 1:
        if (col + n >= msa->length)
        {
          pll_amd_set_error(PLL_ERROR_PHYLIP_LONGSEQ, "Sequence %d (%.100s) longer than expected",
                            row + 1, msa->label[row]);
          return -1;
        }
        dst[n++] = c;
        break;
      case 2:
        report_fatal_char(0, c, fd->lineno);
        return -1;
      default:
        break;
    }
  }
  return n;
}

static pll_msa_t * phy_new_msa(pll_phylip_t * fd)
{
  pll_msa_t * msa = (pll_msa_t *)malloc(sizeof(pll_msa_t));
  if (!msa)
  {
    out_of_memory();
    return NULL;
  }
  if (!phy_header(fd->line, &msa->count, &msa->length))
  {
    free(msa);
    return NULL;
  }
  msa->sequence = (char **)calloc((size_t)msa->count, sizeof(char *));
  msa->label = (char **)calloc((size_t)msa->count, sizeof(char *));
  int ok = msa->sequence && msa->label;
  for (int i = 0; ok && i < msa->count; ++i)
  {
    msa->sequence[i] = (char *)malloc((size_t)msa->length + 1);
    if (msa->sequence[i]) msa->sequence[i][msa->length] = 0;
    else ok = 0;
  }
  if (!ok)
  {
    out_of_memory();
    pll_msa_destroy(msa);
    return NULL;
  }
  return msa;
}

/* The label of row `row` from the start of *p: it ends at the first space of the
 * line if the line has one, else at the first tab, else CR, else LF.  Advances *p
 * past it. */
static int phy_label(pll_msa_t * msa, int row, char ** p)
{
  static const char stops[] = {' ', '\t', '\r', '\n'};
  int stop = '\n';
  for (unsigned int i = 0; i < sizeof(stops); ++i)
    if (strchr(*p, stops[i]))
    {
      stop = stops[i];
      break;
    }
  const long len = span_until(*p, stop);
  assert(len > 0);
  msa->label[row] = (char *)malloc((size_t)len + 1);
  if (!msa->label[row])
  {
    out_of_memory();
    return 0;
  }
  memcpy(msa->label[row], *p, (size_t)len);
  msa->label[row][len] = 0;
  *p += len;
  return 1;
}

/* Interleaved blocks: take the data of the first line, starting with `text`,
 * that holds any; every row of a block must contribute `*block_len` columns.
 * Returns 1 when a row was read, 0 at end of file, -1 on error. */
static int phy_block_row(pll_phylip_t * fd, pll_msa_t * msa, char * text, int row, int col,
                         int * block_len)
{
  while (text)
  {
    const int n = phy_take(fd, msa, text, row, col);
    if (n < 0) return -1;
    if (n)
    {
      if (!*block_len) *block_len = n;
      else if (*block_len != n)
      {
        pll_amd_set_error(PLL_ERROR_PHYLIP_NONALIGNED, "Sequence %d (%.100s) data out of alignment",
                          row + 1, msa->label[row]);
        return -1;
      }
      return 1;
    }
    text = phy_next_line(fd);
  }
  return 0;
}

static pll_msa_t * phy_fail(pll_msa_t * msa)
{
  pll_msa_destroy(msa);
  return NULL;
}

pll_msa_t * pll_phylip_parse_interleaved(pll_phylip_t * fd)
{
  pll_msa_t * msa = phy_new_msa(fd);
  if (!msa) return NULL;

  /* first block: label + data for every row */
  int row = 0, block_len = 0, status = 1;
  char * p;
  while (row < msa->count && (p = phy_next_line(fd)))
  {
    while (*p && is_blank(*p)) ++p;
    if (!*p) continue;
    if (!phy_label(msa, row, &p)) return phy_fail(msa);
    status = phy_block_row(fd, msa, p, row, 0, &block_len);
    if (status <= 0) break;
    ++row;
  }
  if (status < 0) return phy_fail(msa);
  if (row != msa->count)
  {
    pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Found %d sequence(s) but expected %d", row, msa->count);
    return phy_fail(msa);
  }

  /* remaining blocks: data only, rows in the same order */
  int done = block_len, block = 2;
  row = 0;
  block_len = 0;
  for (;;)
  {
    status = phy_block_row(fd, msa, phy_next_line(fd), row, done, &block_len);
    if (status <= 0) break;
    if (++row == msa->count)
    {
      row = 0;
      done += block_len;
      block_len = 0;
      ++block;
    }
  }
  if (status < 0) return phy_fail(msa);
  if (row)
  {
    pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Found %d sequences in block %d but expected %d", row,
                      block, msa->count);
    return phy_fail(msa);
  }
  if (done != msa->length)
  {
    /* (message only: the reference leaves pll_errno as it was) */
    snprintf(pll_errmsg, sizeof(pll_errmsg), "Sequence length is %d but expected %d", done, msa->length);
    return phy_fail(msa);
  }
  return msa;
}

pll_msa_t * pll_phylip_parse_sequential(pll_phylip_t * fd)
{
  pll_msa_t * msa = phy_new_msa(fd);
  if (!msa) return NULL;

  int row = 0;
  char * p;
  while ((p = phy_next_line(fd)))
  {
    while (*p && is_blank(*p)) ++p;
    if (!*p) continue;
    if (row == msa->count)
    {
      pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Found at least %d sequences but expected %d", row + 1,
                        msa->count);
      return phy_fail(msa);
    }
    if (!phy_label(msa, row, &p)) return phy_fail(msa);
    /* the row's data may run over any number of lines */
    for (int col = 0;;)
    {
      const int n = phy_take(fd, msa, p, row, col);
      if (n < 0) return phy_fail(msa);
      col += n;
      if (col == msa->length) break;
      if (!(p = phy_next_line(fd)))
      {
        pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Sequence %d (%.100s) has %d characters but expected %d",
                          row + 1, msa->label[row], col, msa->length);
        return phy_fail(msa);
      }
    }
    ++row;
  }
  if (row != msa->count)
  {
    pll_amd_set_error(PLL_ERROR_PHYLIP_SYNTAX, "Found %d sequence(s) but expected %d", row, msa->count);
    return phy_fail(msa);
  }
  return msa;
}
