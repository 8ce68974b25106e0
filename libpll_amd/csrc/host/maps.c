/* maps.c -- character -> state-bitmask tables for the tip encoders.
 *
 * Replaces pll_map_bin / pll_map_nt / pll_map_aa (maps.c:26,46,66 of the
 * reference).  Bit i of an entry = state i is compatible with the character;
 * 0 = illegal character.  DNA follows the IUPAC ambiguity codes with state order
 * A,C,G,T; amino acids use the order ARNDCQEGHILKMFPSTWYV with B = N|D,
 * Z = Q|E and X/gap = anything.
 */
#include "pll_amd.h"

#define BOTH(c, v) [c] = (v), [(c) + 32] = (v) /* upper and lower case letter */

const unsigned int pll_map_bin[256] = {
  ['0'] = 1, ['1'] = 2, ['-'] = 3, ['?'] = 3,
};

const unsigned int pll_map_nt[256] = {
  BOTH('A', 1),  BOTH('C', 2),  BOTH('G', 4),  BOTH('T', 8),  BOTH('U', 8),
  BOTH('M', 3),  BOTH('R', 5),  BOTH('S', 6),  BOTH('V', 7),  BOTH('W', 9),
  BOTH('Y', 10), BOTH('H', 11), BOTH('K', 12), BOTH('D', 13), BOTH('B', 14),
  BOTH('N', 15), BOTH('O', 15), BOTH('X', 15), ['-'] = 15, ['?'] = 15,
};

#define AA(i) (1u << (i))
#define AA_ANY 0xFFFFFu

const unsigned int pll_map_aa[256] = {
  BOTH('A', AA(0)),  BOTH('R', AA(1)),  BOTH('N', AA(2)),  BOTH('D', AA(3)),
  BOTH('C', AA(4)),  BOTH('Q', AA(5)),  BOTH('E', AA(6)),  BOTH('G', AA(7)),
  BOTH('H', AA(8)),  BOTH('I', AA(9)),  BOTH('L', AA(10)), BOTH('K', AA(11)),
  BOTH('M', AA(12)), BOTH('F', AA(13)), BOTH('P', AA(14)), BOTH('S', AA(15)),
  BOTH('T', AA(16)), BOTH('W', AA(17)), BOTH('Y', AA(18)), BOTH('V', AA(19)),
  BOTH('B', AA(2) | AA(3)), BOTH('Z', AA(5) | AA(6)), BOTH('X', AA_ANY),
  ['*'] = AA_ANY, ['-'] = AA_ANY, ['?'] = AA_ANY,
};
