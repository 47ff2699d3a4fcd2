/* Test infrastructure (oracle side): unpacks a DBoW3 binary vocabulary into its raw, uncompressed stream.
 * The container format is DBoW3's Vocabulary::toStream / fromStream (third_party/DBow3/src/Vocabulary.cpp:1182-1411
 * of the reference): u64 signature 88877711233, u8 compressed, u32 nnodes, then -- if compressed -- u32 nChunks
 * and nChunks QuickLZ blocks (9-byte header carrying the compressed size).  The QuickLZ decoder is the
 * reference's own third_party/DBow3/src/quicklz.c, compiled from where it lies by oracle/ref_build/build_unpack.py;
 * this driver is ours.  Output: u32 nnodes followed by the uncompressed stream (k, L, scoring, weighting, nodes, words).
 * usage: unpack_vocab <vocabulary file> <out.raw> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "quicklz.h"

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in out\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb"), *o = fopen(argv[2], "wb");
  if (!f || !o) { perror("open"); return 1; }
  uint64_t sig = 0; uint8_t comp = 0; uint32_t nnodes = 0, nchunks = 0;
  if (fread(&sig, 8, 1, f) != 1 || sig != 88877711233ULL) { fprintf(stderr, "not a DBoW3 binary vocabulary\n"); return 1; }
  if (fread(&comp, 1, 1, f) != 1 || fread(&nnodes, 4, 1, f) != 1) return 1;
  fwrite(&nnodes, 4, 1, o);
  if (!comp) {
    char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) fwrite(buf, 1, n, o);
  } else {
    qlz_state_decompress *st = calloc(1, sizeof *st);
    char *in = malloc(10000 + 400), *out = malloc(10000 + 400);
    if (fread(&nchunks, 4, 1, f) != 1) return 1;
    for (uint32_t i = 0; i < nchunks; ++i) {
      if (fread(in, 1, 9, f) != 9) return 1;
      size_t c = qlz_size_compressed(in);
      if (c < 9 || c > 10400 || fread(in + 9, 1, c - 9, f) != c - 9) return 1;
      size_t d = qlz_decompress(in, out, st);
      fwrite(out, 1, d, o);
    }
  }
  fclose(o); fclose(f);
  return 0;
}
