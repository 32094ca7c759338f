/* Drop-in header for the Turing encoder's `#include "havoc/hadamard.h"` (turing/StateFunctionTables.h:26-33):
 * every declaration lives in havoc_tables.hpp; the implementation is libhavoc_classic.so over libhavoc_mi355x.so. */
#include "havoc_tables.hpp"
