/* Path shim: the reference host code includes "beatricelib/beatrice.h" (e.g. reference
 * src/common/processor_core_2.cc:16).  Pointing its include path at this repo's include/
 * resolves that name to our declaration of the same C-ABI. */
#ifndef BEATRICE_H_
#define BEATRICE_H_
#include "../beatrice_abi.h"
#endif
