/* gen_A: build the Jacobian-preconditioner matrix file from a circulation file.
 *
 *    gen_A [-h] [-D dbg_lvl] [-o opt_fname] matrix_fname
 *
 * Drop-in for the reference's generator front end (src/gen_A.c:27-501): same command line, same option-file
 * language, same defaults, same report at -D1 and the same exit codes and diagnostics a job script can grep for.
 * Written from that grammar as data: one descriptor per option key (how many values, what they are, where they
 * go) and one per sink flavour; the parser itself knows no option by name.
 *
 * option file: one "key value [value ...]" per line, at most 255 characters:
 *    day_cnt <days>                      reg_fname <file>         circ_fname <file>
 *    adv_type none|donor|cent...|upwind3 l_adv_enforce_divfree 0|1
 *    hmix_type none|const|hor_file|isop_file
 *    vmix_type none|const|file|matrix_file
 *    tracer_fname <file>                 coupled_tracer_cnt 1|2   tracer_ind <n>
 *    sink_type none | const <rate> | const_shallow <rate> <depth> | file <field>
 *              | generic_tracer <name> [depends_layer_cnt]        (applies to tracer_ind)
 *    pv <field>                          sf <field>               (apply to tracer_ind)
 *    coupled_tracer_type none|OCMIP_BGC_PO4_DOP|DIC_SHADOW_ALK_SHADOW
 *
 * Pure host program: no GPU is involved in generating the file.
 */
#include <getopt.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../host/nkp_host.h"

#define LINE_CAP 256
#define MAX_WORDS 8

static const char *USAGE = "usage: gen_matrix_file [-h] [-D dbg_lvl] [-o opt_fname] matrix_fname";

static struct {
   char *opt_file, *matrix_file;
   double day_cnt;
   int tracer;                   /* the tracer that sink_type / pv / sf lines currently address */
} job = { NULL, NULL, 365.0, 0 };

/* ---------------------------------------------------------------- vocabularies (enum code <-> word) */
typedef struct { const char *word; int code; int prefix; } vocab;      /* prefix > 0: that many leading characters decide */

static const vocab ADV[] = { { "none", adv_none, 0 }, { "donor", adv_donor, 0 }, { "centered", adv_cent, 4 }, { "upwind3", adv_upwind3, 0 }, { NULL, 0, 0 } };
static const vocab HMIX[] = { { "none", hmix_none, 0 }, { "const", hmix_const, 0 }, { "hor_file", hmix_hor_file, 0 }, { "isop_file", hmix_isop_file, 0 }, { NULL, 0, 0 } };
static const vocab VMIX[] = { { "none", vmix_none, 0 }, { "const", vmix_const, 0 }, { "file", vmix_file, 0 }, { "matrix_file", vmix_matrix_file, 0 }, { NULL, 0, 0 } };
static const vocab COUPLING[] = { { "none", coupled_tracer_none, 0 }, { "OCMIP_BGC_PO4_DOP", coupled_tracer_OCMIP_BGC_PO4_DOP, 0 },
                                  { "DIC_SHADOW_ALK_SHADOW", coupled_tracer_DIC_SHADOW_ALK_SHADOW, 0 }, { NULL, 0, 0 } };
static const vocab SWITCH01[] = { { "0", 0, 0 }, { "1", 1, 0 }, { NULL, 0, 0 } };

static const vocab *vocab_find (const vocab *v, const char *word)
{
   for (; v->word; v++)
      if (v->prefix ? strncmp (v->word, word, (size_t) v->prefix) == 0 : strcmp (v->word, word) == 0) return v;
   return NULL;
}

static const char *vocab_word (const vocab *v, int code)
{
   for (; v->word; v++)
      if (v->code == code) return v->word;
   return "?";
}

/* ---------------------------------------------------------------- per-tracer records */
static const per_tracer_opt_t TRACER_DEFAULTS = {
   .sink_opt = sink_none, .sink_rate = 1.21e-4 /* radiocarbon decay */, .sink_depth = 10.0e2 /* 10 m */,
   .sink_field_name = NULL, .sink_generic_tracer_name = NULL, .sink_generic_tracer_depends_layer_cnt = -1,
   .pv_field_name = NULL, .d_SF_d_TRACER_field_name = NULL,
};

static int resize_tracer_records (int have, int want)
{
   per_tracer_opt_t *p = want >= 1 ? (per_tracer_opt_t *) realloc (per_tracer_opt, (size_t) want * sizeof *p) : NULL;
   if (p == NULL) {
      fprintf (stderr, "(%d) realloc failed in grow_per_tracer_opt for grow_per_tracer_opt\n", iam);
      return 1;
   }
   per_tracer_opt = p;
   for (int t = have; t < want; t++) per_tracer_opt[t] = TRACER_DEFAULTS;
   return 0;
}

/* ---------------------------------------------------------------- value setters shared by all descriptors */
static int keep_text (const char *what, const char *value, char **slot)
{
   char *copy = (char *) malloc (strlen (value) + 1);
   if (copy == NULL) {
      fprintf (stderr, "(%d) malloc failed in read_opt_file for %s\n", iam, what);
      return 1;
   }
   strcpy (copy, value);
   *slot = copy;
   return 0;
}

static int bad_number (const char *value, const char *key)
{
   fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, value, key);
   return 1;
}

/* ---------------------------------------------------------------- sink flavours: what follows the flavour word */
enum arg_kind { ARG_RATE, ARG_DEPTH, ARG_FIELD, ARG_GENERIC_NAME, ARG_LAYER_CNT };
typedef struct { enum arg_kind kind; const char *label; int optional; } sink_arg;
typedef struct { const char *word; sink_opt_t code; int nargs; sink_arg args[2]; } sink_flavour;

static const sink_flavour SINKS[] = {
   { "none", sink_none, 0, { { 0, NULL, 0 }, { 0, NULL, 0 } } },
   { "const", sink_const, 1, { { ARG_RATE, "sink_rate", 0 }, { 0, NULL, 0 } } },
   { "const_shallow", sink_const_shallow, 2, { { ARG_RATE, "sink_rate", 0 }, { ARG_DEPTH, "sink_depth", 0 } } },
   { "file", sink_file, 1, { { ARG_FIELD, "sink_field_name", 0 }, { 0, NULL, 0 } } },
   { "generic_tracer", sink_generic_tracer, 2, { { ARG_GENERIC_NAME, "sink_generic_tracer_name", 0 }, { ARG_LAYER_CNT, "sink_generic_tracer_depends_layer_cnt", 1 } } },
};
#define N_SINKS ((int) (sizeof SINKS / sizeof SINKS[0]))

static int take_sink (const char *key, char **w, int nw)
{
   per_tracer_opt_t *rec = &per_tracer_opt[job.tracer];
   const sink_flavour *fl = NULL;

   for (int e = 0; e < N_SINKS && !fl; e++)
      if (strcmp (SINKS[e].word, w[0]) == 0) fl = &SINKS[e];
   if (!fl) {
      fprintf (stderr, "(%d) unknown %s: %s\n", iam, key, w[0]);
      return 1;
   }
   rec->sink_opt = fl->code;
   for (int a = 0; a < fl->nargs; a++) {
      const sink_arg *sa = &fl->args[a];
      const char *v = (a + 1 < nw) ? w[a + 1] : NULL;
      if (v == NULL) {
         if (sa->optional) break;
         fprintf (stderr, "(%d) unspecified %s\n", iam, sa->label);
         return 1;
      }
      switch (sa->kind) {
      case ARG_RATE: if (parse_to_double ((char *) v, &rec->sink_rate)) return bad_number (v, key); break;
      case ARG_DEPTH: if (parse_to_double ((char *) v, &rec->sink_depth)) return bad_number (v, key); break;
      case ARG_FIELD: if (keep_text (sa->label, v, &rec->sink_field_name)) return 1; break;
      case ARG_GENERIC_NAME: if (keep_text (sa->label, v, &rec->sink_generic_tracer_name)) return 1; break;
      case ARG_LAYER_CNT:
         if (parse_to_int ((char *) v, &rec->sink_generic_tracer_depends_layer_cnt)) {
            fprintf (stderr, "(%d) error parsing %s\n", iam, sa->label);
            return 1;
         }
         break;
      }
   }
   return 0;
}

/* ---------------------------------------------------------------- the option keys */
enum key_kind { K_REAL, K_TEXT, K_WORD, K_TRACER_TEXT, K_TRACER_COUNT, K_TRACER_SELECT, K_SINK };
typedef struct {
   const char *key;
   enum key_kind kind;
   void *slot;                   /* K_REAL: double *, K_TEXT: char **, K_WORD: int * */
   const vocab *words;           /* K_WORD */
   size_t member;                /* K_TRACER_TEXT: offset of the char * inside per_tracer_opt_t */
   const char *label;            /* name used in allocation diagnostics */
} key_desc;

static int adv_code, hmix_code, vmix_code, coupling_code;     /* enum-typed globals are set from these after parsing */

static const key_desc KEYS[] = {
   { "day_cnt", K_REAL, &job.day_cnt, NULL, 0, NULL },
   { "reg_fname", K_TEXT, &reg_fname, NULL, 0, "reg_fname" },
   { "circ_fname", K_TEXT, &circ_fname, NULL, 0, "circ_fname" },
   { "tracer_fname", K_TEXT, &tracer_fname, NULL, 0, "tracer_fname" },
   { "adv_type", K_WORD, &adv_code, ADV, 0, NULL },
   { "l_adv_enforce_divfree", K_WORD, &l_adv_enforce_divfree, SWITCH01, 0, NULL },
   { "hmix_type", K_WORD, &hmix_code, HMIX, 0, NULL },
   { "vmix_type", K_WORD, &vmix_code, VMIX, 0, NULL },
   { "coupled_tracer_type", K_WORD, &coupling_code, COUPLING, 0, NULL },
   { "coupled_tracer_cnt", K_TRACER_COUNT, NULL, NULL, 0, NULL },
   { "tracer_ind", K_TRACER_SELECT, NULL, NULL, 0, NULL },
   { "sink_type", K_SINK, NULL, NULL, 0, NULL },
   { "pv", K_TRACER_TEXT, NULL, NULL, offsetof (per_tracer_opt_t, pv_field_name), "pv_field_name" },
   { "sf", K_TRACER_TEXT, NULL, NULL, offsetof (per_tracer_opt_t, d_SF_d_TRACER_field_name), "d_SF_d_TRACER_field_name" },
};
#define N_KEYS ((int) (sizeof KEYS / sizeof KEYS[0]))

static int take_option (const key_desc *k, char **w, int nw)
{
   int n;

   switch (k->kind) {
   case K_REAL:
      return parse_to_double (w[0], (double *) k->slot) ? bad_number (w[0], k->key) : 0;
   case K_TEXT:
      return keep_text (k->label, w[0], (char **) k->slot);
   case K_TRACER_TEXT:
      return keep_text (k->label, w[0], (char **) ((char *) &per_tracer_opt[job.tracer] + k->member));
   case K_WORD: {
      const vocab *v = vocab_find (k->words, w[0]);
      if (!v) {
         fprintf (stderr, "(%d) unknown %s: %s\n", iam, k->key, w[0]);
         return 1;
      }
      *(int *) k->slot = v->code;
      return 0;
   }
   case K_TRACER_COUNT:
      if (parse_to_int (w[0], &n)) return bad_number (w[0], k->key);
      if (n < 1 || n > 2) {                       /* checked before any record is allocated */
         fprintf (stderr, "(%d) coupled_tracer_cnt = %d not supported\n", iam, n);
         return 1;
      }
      if (resize_tracer_records (coupled_tracer_cnt, n)) {
         fprintf (stderr, "(%d) error from grow_per_tracer_opt\n", iam);
         return 1;
      }
      coupled_tracer_cnt = n;
      return 0;
   case K_TRACER_SELECT:
      if (parse_to_int (w[0], &n)) return bad_number (w[0], k->key);
      if (n < 0 || n >= coupled_tracer_cnt) {
         fprintf (stderr, "(%d) tracer_ind = %d out of bounds for coupled_tracer_cnt = %d\n", iam, n, coupled_tracer_cnt);
         return 1;
      }
      job.tracer = n;
      return 0;
   case K_SINK:
      return take_sink (k->key, w, nw);
   }
   return 1;
}

/* split a line in place at blanks; returns the number of words (at most MAX_WORDS are kept) */
static int split_words (char *line, char **w)
{
   int nw = 0;
   for (char *p = line; *p;) {
      while (*p == ' ' || *p == '\n') *p++ = '\0';
      if (!*p) break;
      if (nw < MAX_WORDS) w[nw++] = p;
      while (*p && *p != ' ' && *p != '\n') p++;
   }
   return nw;
}

static int read_option_file (void)
{
   char line[LINE_CAP], *w[MAX_WORDS];
   int lineno = 0;
   FILE *fp;

   if (job.opt_file == NULL) return 0;
   if ((fp = fopen (job.opt_file, "r")) == NULL) {
      fprintf (stderr, "(%d) fopen failed in read_opt_file for %s\n", iam, job.opt_file);
      return 1;
   }
   while (fgets (line, LINE_CAP, fp)) {
      const size_t len = strlen (line);
      const key_desc *k = NULL;

      lineno++;
      if (len == 0) continue;
      if (line[len - 1] != '\n') {
         fprintf (stderr, "(%d) line number %d in %s too long\n", iam, lineno, job.opt_file);
         fclose (fp);
         return 1;
      }
      const int nw = split_words (line, w);
      if (nw == 0) continue;                      /* a blank line is legal here (the reference crashes on one) */
      if (nw == 1) {
         fprintf (stderr, "(%d) unspecified value for %s\n", iam, w[0]);
         fclose (fp);
         return 1;
      }
      for (int e = 0; e < N_KEYS && !k; e++)
         if (strcmp (KEYS[e].key, w[0]) == 0) k = &KEYS[e];
      if (!k) {
         fprintf (stderr, "(%d) unknown option name: %s\n", iam, w[0]);
         fclose (fp);
         return 1;
      }
      if (take_option (k, w + 1, nw - 1)) {
         fclose (fp);
         return 1;
      }
   }
   fclose (fp);
   return 0;
}

/* ---------------------------------------------------------------- -D1 report */
static void report (void)
{
   if (!dbg_lvl) return;
#define SHOW(label, fmt, value) printf ("(%d) %-26s = " fmt "\n", iam, label, value)
#define SHOW_T(label, fmt, value) printf ("(%d)    %-23s = " fmt "\n", iam, label, value)
   SHOW ("dbg_lvl", "%d", dbg_lvl);
   SHOW ("day_cnt", "%e", job.day_cnt);
   SHOW ("reg_fname", "%s", reg_fname ? reg_fname : "none");
   SHOW ("circ_fname", "%s", circ_fname);
   SHOW ("adv_opt", "%s", vocab_word (ADV, adv_opt));
   SHOW ("l_adv_enforce_divfree", "%d", l_adv_enforce_divfree);
   SHOW ("hmix_opt", "%s", vocab_word (HMIX, hmix_opt));
   SHOW ("vmix_opt", "%s", vocab_word (VMIX, vmix_opt));
   SHOW ("tracer_fname", "%s", tracer_fname ? tracer_fname : "none");
   SHOW ("coupled_tracer_cnt", "%d", coupled_tracer_cnt);
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      const per_tracer_opt_t *rec = &per_tracer_opt[t];
      const sink_flavour *fl = &SINKS[0];

      for (int e = 0; e < N_SINKS; e++)
         if (SINKS[e].code == rec->sink_opt) fl = &SINKS[e];
      printf ("(%d) options for tracer %d\n", iam, t);
      SHOW_T ("sink_opt", "%s", fl->word);
      for (int a = 0; a < fl->nargs; a++)
         switch (fl->args[a].kind) {
         case ARG_RATE: SHOW_T ("sink_rate", "%e", rec->sink_rate); break;
         case ARG_DEPTH: SHOW_T ("sink_depth", "%e", rec->sink_depth); break;
         case ARG_FIELD: SHOW_T ("sink_field_name", "%s", rec->sink_field_name); break;
         case ARG_GENERIC_NAME: printf ("(%d)    sink_generic_tracer_name= %s\n", iam, rec->sink_generic_tracer_name); break;
         case ARG_LAYER_CNT: SHOW_T ("depends_layer_cnt", "%d", rec->sink_generic_tracer_depends_layer_cnt); break;
         }
      SHOW_T ("pv_field_name", "%s", rec->pv_field_name ? rec->pv_field_name : "none");
      printf ("(%d)    d_SF_d_TRACER_field_name= %s\n", iam, rec->d_SF_d_TRACER_field_name ? rec->d_SF_d_TRACER_field_name : "none");
   }
   SHOW ("coupled_tracer_opt", "%s", vocab_word (COUPLING, coupled_tracer_opt));
   printf ("(%d) %-26s = %s\n\n", iam, "matrix_fname", job.matrix_file);
#undef SHOW
#undef SHOW_T
}

/* ---------------------------------------------------------------- command line */
static int read_command_line (int argc, char **argv)
{
   int c;

   while ((c = getopt (argc, argv, "D:o:h")) != -1) {
      if (c == 'D') {
         if (parse_to_int (optarg, &dbg_lvl)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%c'\n", iam, optarg, c);
            return 1;
         }
      } else if (c == 'o')
         job.opt_file = optarg;
      else {                                       /* -h and anything getopt rejects */
         fprintf (stderr, "(%d) %s\n", iam, USAGE);
         return 1;
      }
   }
   if (argc - optind != 1) {
      fprintf (stderr, "(%d) unexpected number of arguments\n%s\n", iam, USAGE);
      return 1;
   }
   job.matrix_file = argv[optind];
   return 0;
}

int main (int argc, char *argv[])
{
   iam = 0;
   dbg_lvl = 0;
   if (read_command_line (argc, argv)) return EXIT_FAILURE;

   /* defaults of the reference (src/gen_A.c:96-110): one year, centred advection made divergence-free, isopycnal
    * mixing and vertical mixing from the circulation file, a single tracer without sink */
   adv_code = adv_cent;
   hmix_code = hmix_isop_file;
   vmix_code = vmix_file;
   coupling_code = coupled_tracer_none;
   l_adv_enforce_divfree = 1;
   coupled_tracer_cnt = 1;
   if (resize_tracer_records (0, 1)) {
      fprintf (stderr, "(%d) error from grow_per_tracer_opt\n", iam);
      return EXIT_FAILURE;
   }
   if (read_option_file ()) return EXIT_FAILURE;
   adv_opt = (adv_opt_t) adv_code;
   hmix_opt = (hmix_opt_t) hmix_code;
   vmix_opt = (vmix_opt_t) vmix_code;
   coupled_tracer_opt = (coupled_tracer_opt_t) coupling_code;
   if (coupled_tracer_cnt == 2 && coupled_tracer_opt == coupled_tracer_none) {
      fprintf (stderr, "(%d) coupled_tracer_cnt = 2 only supported for coupled_tracer_type = OCMIP_BGC_PO4_DOP, DIC_SHADOW_ALK_SHADOW\n", iam);
      return EXIT_FAILURE;
   }
   if (circ_fname == NULL) {                       /* the reference goes on to open a NULL file name */
      fprintf (stderr, "(%d) circ_fname not specified (option file line: circ_fname <file>)\n", iam);
      return EXIT_FAILURE;
   }
   report ();

   /* grid section, index maps, matrix: each stage is generated, then appended to the matrix file */
   if (get_grid_info (circ_fname, reg_fname) || put_grid_info (job.matrix_file)) return EXIT_FAILURE;
   if (gen_ind_maps () || put_ind_maps (job.matrix_file)) return EXIT_FAILURE;
   if (gen_sparse_matrix (job.day_cnt) || put_sparse_matrix (job.matrix_file)) return EXIT_FAILURE;

   free_sparse_matrix ();
   free_ind_maps ();
   free_grid_info ();
   free (per_tracer_opt);
   return EXIT_SUCCESS;
}
