/* tree.c -- from a tree to the op list the hot path consumes.
 *
 * Replaces pll_utree_traverse (utree.c:403), pll_utree_create_operations
 * (utree.c:284), pll_rtree_traverse (rtree.c:361) and
 * pll_rtree_create_operations (rtree.c:262): the two calls every client makes
 * right before pll_update_prob_matrices / pll_update_partials.  Parsers, tree
 * surgery and export stay out of scope; clients (and our harness) build the
 * node graphs themselves, as the reference's partial-traversal test does.
 *
 * Host pointer-chasing, no arithmetic.  The reference recurses; these walk with
 * an explicit stack so a 100 000-tip caterpillar cannot overflow the C stack.
 * Visit order, callback protocol and outputs are the reference's.
 */
#include <stdio.h>

#include "internal.h"

typedef struct
{
  void * node;
  int stage; /* 0 = not yet expanded, 1 = children pushed (post-order emit pending) */
} frame_t;

typedef struct
{
  frame_t * f;
  size_t n, cap;
} stack_t;

static int push(stack_t * s, void * node, int stage)
{
  if (s->n == s->cap)
  {
    size_t cap = s->cap ? 2 * s->cap : 256;
    frame_t * f = (frame_t *)realloc(s->f, cap * sizeof(frame_t));
    if (!f) return 0;
    s->f = f;
    s->cap = cap;
  }
  s->f[s->n].node = node;
  s->f[s->n].stage = stage;
  s->n++;
  return 1;
}

/* generic walk: child accessors abstract over the two node types */
typedef void * (*child_fn)(void * node, int which);
typedef int (*is_tip_fn)(void * node);

static int walk(void * start, int postorder, int (*cb)(void *), child_fn child, is_tip_fn is_tip,
                void ** out, unsigned int * count)
{
  stack_t s = {NULL, 0, 0};
  if (!push(&s, start, 0)) goto oom;
  while (s.n)
  {
    frame_t fr = s.f[--s.n];
    if (fr.stage == 1)
    {
      out[(*count)++] = fr.node;
      continue;
    }
    if (is_tip(fr.node))
    {
      /* a tip is reported iff the callback accepts it (utree.c:381-388) */
      if (cb(fr.node)) out[(*count)++] = fr.node;
      continue;
    }
    /* an inner node the callback rejects prunes its whole subtree (utree.c:390) */
    if (!cb(fr.node)) continue;
    if (postorder)
    {
      if (!push(&s, fr.node, 1)) goto oom;
    }
    else
      out[(*count)++] = fr.node;
    /* first child must be processed first: push it last */
    if (!push(&s, child(fr.node, 1), 0) || !push(&s, child(fr.node, 0), 0)) goto oom;
  }
  free(s.f);
  return PLL_SUCCESS;
oom:
  free(s.f);
  pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
  return PLL_FAILURE;
}

/* ---- unrooted: an inner node is a ring of three pll_unode_t ---- */

static void * uchild(void * n, int which)
{
  pll_unode_t * u = (pll_unode_t *)n;
  return which == 0 ? (void *)u->next->back : (void *)u->next->next->back;
}

static int utip(void * n) { return ((pll_unode_t *)n)->next == NULL; }

int pll_utree_traverse(pll_unode_t * root, int traversal, int (*cbtrav)(pll_unode_t *),
                       pll_unode_t ** outbuffer, unsigned int * trav_size)
{
  int post;
  *trav_size = 0;
  if (!root->next) return PLL_FAILURE;
  if (traversal == PLL_TREE_TRAVERSE_POSTORDER)
    post = 1;
  else if (traversal == PLL_TREE_TRAVERSE_PREORDER)
    post = 0;
  else
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Invalid traversal value.");
    return PLL_FAILURE;
  }
  /* both sides of the root edge: first the subtree behind root->back, then root's */
  if (!walk(root->back, post, (int (*)(void *))cbtrav, uchild, utip, (void **)outbuffer, trav_size))
    return PLL_FAILURE;
  return walk(root, post, (int (*)(void *))cbtrav, uchild, utip, (void **)outbuffer, trav_size);
}

void pll_utree_create_operations(pll_unode_t * const * trav, unsigned int trav_size,
                                 double * branches, unsigned int * pmatrix_indices,
                                 pll_operation_t * ops, unsigned int * matrix_count,
                                 unsigned int * ops_count)
{
  unsigned int i;
  const pll_unode_t * skip = trav_size ? trav[trav_size - 1]->back : NULL;
  *ops_count = 0;
  *matrix_count = 0;
  for (i = 0; i < trav_size; ++i)
  {
    const pll_unode_t * n = trav[i];
    /* the root edge has two end points in the buffer: list it once (utree.c:305-314) */
    if (n != skip)
    {
      branches[*matrix_count] = n->length;
      pmatrix_indices[*matrix_count] = n->pmatrix_index;
      ++*matrix_count;
    }
    if (n->next)
    {
      const pll_unode_t * a = n->next->back, * b = n->next->next->back;
      pll_operation_t * op = ops + (*ops_count)++;
      op->parent_clv_index = n->clv_index;
      op->parent_scaler_index = n->scaler_index;
      op->child1_clv_index = a->clv_index;
      op->child1_scaler_index = a->scaler_index;
      op->child1_matrix_index = a->pmatrix_index;
      op->child2_clv_index = b->clv_index;
      op->child2_scaler_index = b->scaler_index;
      op->child2_matrix_index = b->pmatrix_index;
    }
  }
}

/* ---- rooted ---- */

static void * rchild(void * n, int which)
{
  pll_rnode_t * r = (pll_rnode_t *)n;
  return which == 0 ? (void *)r->left : (void *)r->right;
}

static int rtip(void * n) { return ((pll_rnode_t *)n)->left == NULL; }

int pll_rtree_traverse(pll_rnode_t * root, int traversal, int (*cbtrav)(pll_rnode_t *),
                       pll_rnode_t ** outbuffer, unsigned int * trav_size)
{
  int post;
  *trav_size = 0;
  if (!root->left) return PLL_FAILURE;
  if (traversal == PLL_TREE_TRAVERSE_POSTORDER)
    post = 1;
  else if (traversal == PLL_TREE_TRAVERSE_PREORDER)
    post = 0;
  else
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Invalid traversal value.");
    return PLL_FAILURE;
  }
  return walk(root, post, (int (*)(void *))cbtrav, rchild, rtip, (void **)outbuffer, trav_size);
}

void pll_rtree_create_operations(pll_rnode_t * const * trav, unsigned int trav_size,
                                 double * branches, unsigned int * pmatrix_indices,
                                 pll_operation_t * ops, unsigned int * matrix_count,
                                 unsigned int * ops_count)
{
  unsigned int i;
  *ops_count = 0;
  *matrix_count = 0;
  for (i = 0; i < trav_size; ++i)
  {
    const pll_rnode_t * n = trav[i];
    /* the last node is the root: it has no branch (rtree.c:283-289) */
    if (i + 1 < trav_size)
    {
      branches[*matrix_count] = n->length;
      pmatrix_indices[*matrix_count] = n->pmatrix_index;
      ++*matrix_count;
    }
    if (n->left)
    {
      pll_operation_t * op = ops + (*ops_count)++;
      op->parent_clv_index = n->clv_index;
      op->parent_scaler_index = n->scaler_index;
      op->child1_clv_index = n->left->clv_index;
      op->child1_scaler_index = n->left->scaler_index;
      op->child1_matrix_index = n->left->pmatrix_index;
      op->child2_clv_index = n->right->clv_index;
      op->child2_scaler_index = n->right->scaler_index;
      op->child2_matrix_index = n->right->pmatrix_index;
    }
  }
}
