/*
 * checkers_mt.c -- LDSSChecker.Check (LDSSChecker.cs:23-119) evaluated by several threads, for the
 * full-size configurations (256 MiB ... 2 GiB of text) where the sequential walk of checkers.c
 * takes minutes.  Same phases and result codes; see checkers_mt_impl.h.
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "dq_oracle.h"

#define IDX int32_t
#define SUF _i32
#include "checkers_mt_impl.h"
#undef IDX
#undef SUF

#define IDX int64_t
#define SUF _i64
#include "checkers_mt_impl.h"
#undef IDX
#undef SUF
