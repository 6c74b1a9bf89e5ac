/* tmx_expr.h - interpreter of tmx_expr programs (include/tmx.h): the device-evaluable stand-in for the reference's host
 * callbacks sco::ScalarOfVector / sco::VectorOfVector (trajopt_sco/include/trajopt_sco/sco_common.hpp, num_diff.hpp:14-57).
 * ONE header for the kernels, the host front ends and the oracle, so that a user function is the same arithmetic everywhere. */
#ifndef TMX_EXPR_H
#define TMX_EXPR_H
#include "tmx.h"
#include "tmx_detmath.h"
#include <math.h>

/* evaluates the program at x; results in out[0 .. n_outputs).  Returns 0, or -1 for a malformed program (stack under- / overflow,
 * bad index): validate with tmx_expr_check at upload, the kernels do not test the return value. */
TMX_HD static inline int tmx_expr_eval(const int32_t* ops, int32_t n_ops, const double* consts, const double* x, double* out)
{
  double st[TMX_EXPR_STACK];
  int sp = 0;
  for (int32_t k = 0; k < n_ops; ++k)
  {
    const int32_t op = ops[2 * k], arg = ops[2 * k + 1];
    if (op == TMX_OP_VAR || op == TMX_OP_CONST)
    {
      if (sp >= TMX_EXPR_STACK)
        return -1;
      st[sp++] = (op == TMX_OP_VAR) ? x[arg] : consts[arg];
    }
    else if (op == TMX_OP_ADD || op == TMX_OP_SUB || op == TMX_OP_MUL || op == TMX_OP_DIV)
    {
      if (sp < 2)
        return -1;
      const double b = st[--sp], a = st[sp - 1];
      st[sp - 1] = (op == TMX_OP_ADD) ? a + b : (op == TMX_OP_SUB) ? a - b : (op == TMX_OP_MUL) ? a * b : a / b;
    }
    else if (op == TMX_OP_OUT)
    {
      if (sp < 1)
        return -1;
      out[arg] = st[--sp];
    }
    else
    {
      if (sp < 1)
        return -1;
      const double a = st[sp - 1];
      st[sp - 1] = (op == TMX_OP_NEG) ? -a : (op == TMX_OP_SQ) ? a * a : (op == TMX_OP_SIN) ? tmx_sin(a) : (op == TMX_OP_COS) ? tmx_cos(a) : sqrt(a);
    }
  }
  return 0;
}

/* static check of a program against n_vars variables: opcodes, indices, stack discipline, every output written exactly once */
static inline int tmx_expr_check(const tmx_expr* e, int32_t n_vars)
{
  if (!e || e->n_ops < 1 || !e->ops || e->n_outputs < 1 || e->n_outputs > TMX_EXPR_MAX_OUT || e->n_consts < 0 || (e->n_consts > 0 && !e->consts))
    return -1;
  int sp = 0;
  int seen[TMX_EXPR_MAX_OUT] = { 0 };
  for (int32_t k = 0; k < e->n_ops; ++k)
  {
    const int32_t op = e->ops[2 * k], arg = e->ops[2 * k + 1];
    if (op == TMX_OP_VAR || op == TMX_OP_CONST)
    {
      if (arg < 0 || arg >= (op == TMX_OP_VAR ? n_vars : e->n_consts) || sp >= TMX_EXPR_STACK)
        return -1;
      ++sp;
    }
    else if (op == TMX_OP_ADD || op == TMX_OP_SUB || op == TMX_OP_MUL || op == TMX_OP_DIV)
    {
      if (sp < 2)
        return -1;
      --sp;
    }
    else if (op == TMX_OP_OUT)
    {
      if (sp < 1 || arg < 0 || arg >= e->n_outputs || seen[arg])
        return -1;
      seen[arg] = 1;
      --sp;
    }
    else if (op >= TMX_OP_NEG && op <= TMX_OP_SQRT)
    {
      if (sp < 1)
        return -1;
    }
    else
      return -1;
  }
  for (int32_t o = 0; o < e->n_outputs; ++o)
    if (!seen[o])
      return -1;
  return sp == 0 ? 0 : -1;
}
#endif
