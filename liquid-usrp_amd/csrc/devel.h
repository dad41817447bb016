// devel.h -- experiment switches.
//
// The release library's behaviour is a function of its arguments (mcrx_hip_config, the call parameters): no environment
// variable selects a kernel build, a schedule or a buffer count.  Development builds (-DMCRX_DEVEL: `make DEVEL=1`) read the
// experiment switches the measurement scripts under scratch/ use -- MCRX_NSEG, MCRX_SLOTS, MCTX_R8, ... -- through devel_env();
// in a release build devel_env() is a constant nullptr and the branches behind it fold away.
// The one variable a release build reads is MCRX_DEBUG (diagnostic prints on stderr; it changes no result and no schedule).
#pragma once
#include <cstdlib>

#ifdef MCRX_DEVEL
static inline const char *devel_env(const char *name) { return getenv(name); }
#else
static inline const char *devel_env(const char *) { return nullptr; }
#endif
static inline int devel_env_int(const char *name, int dflt) { const char *v = devel_env(name); return v ? atoi(v) : dflt; }
