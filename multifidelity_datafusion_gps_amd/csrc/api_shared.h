// api_shared.h -- the pieces of an evaluation that the C-ABI's translation units share (round 6: mfgp_api.hip was one file for five
// concerns).  Defined in mfgp_api.hip (evaluation) and api_predict.hip (predict); C linkage like the entry points around them --
// they are not part of include/mfgp.h and carry the mfgp_handle by pointer only.
#pragma once
#include "mfgp_internal.h"

#define MFGP_LOCAL __attribute__((visibility("hidden")))   /* shared between the library's translation units, not exported */

extern "C" {
// (re)upload the handle's task list after the planner changed it
MFGP_LOCAL int upload_tasks(mfgp_handle* h);
// One step of a plan.  nbatch > 0: over the handle's batch sets (mfgp_eval_batch) instead of its own slab -- the same launch with one
// more grid dimension; tasks: another task list than the handle's own (a rank's plan of a sharded evaluation).  -> 0, or -1 when
// the planner asked for a kernel that does not exist (h->err says which)
MFGP_LOCAL int run_step(mfgp_handle* h, const mfgp::Step& s, bool want_grad = true, int nbatch = 0, const mfgp::GemmTask* tasks = nullptr);
MFGP_LOCAL float ev_ms(hipEvent_t a, hipEvent_t b);
MFGP_LOCAL int build_plans(mfgp_handle* h);
// release the handle's batch slab (mfgp_eval_batch's matrix sets)
MFGP_LOCAL void free_batch(mfgp_handle* h);
// a handle with data and kernel, else -1 with `who` in the message
MFGP_LOCAL int check_ready(mfgp_handle* h, const char* who);
// validate and store theta / noise / jitter of the evaluation about to be enqueued
MFGP_LOCAL int set_params(mfgp_handle* h, const double* theta, double noise, double jitter);
// K build + sweep + solves (+ gradient) of one evaluation on the handle's streams / its read-back after the synchronisation
MFGP_LOCAL int enqueue_eval(mfgp_handle* h, const double* theta, double noise, double jitter, bool want_grad, bool prebuilt = false);
MFGP_LOCAL int finish_eval(mfgp_handle* h, bool want_grad);
// k(x, x) of the handle's stationary covariance at its current parameters (GPy Kdiag)
MFGP_LOCAL double prior_variance(const mfgp_handle* h);
// make room for a predictive panel of rows_p rows in h->dXs
MFGP_LOCAL int ensure_xs(mfgp_handle* h, int rows_p);
}
