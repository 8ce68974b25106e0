/* pll.h -- drop-in name for clients written against libpll's header: everything
 * this library implements of it lives in pll_amd.h. */
#ifndef PLL_H_
#define PLL_H_
#include "pll_amd.h"
#endif
