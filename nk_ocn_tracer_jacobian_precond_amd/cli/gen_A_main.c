/* gen_A: build the Jacobian-preconditioner matrix file from a circulation file
 * (reference src/gen_A.c:27-501).  Same command line, option-file grammar, defaults,
 * messages and exit codes:
 *
 *    gen_A [-h] [-D dbg_lvl] [-o opt_fname] matrix_fname
 *
 * option file: one "name value [value ...]" per line (at most 255 characters):
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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../host/nkp_host.h"

#define MAX_LINE_LEN 256

static char *opt_fname = NULL;
static double day_cnt;
static char *matrix_fname = NULL;

static int parse_cmd_line (int argc, char **argv)
{
   char *usage_msg = "usage: gen_matrix_file [-h] [-D dbg_lvl] [-o opt_fname] matrix_fname";
   int opt;

   while ((opt = getopt (argc, argv, "D:o:h")) != -1) {
      switch (opt) {
      case '?':
      case 'h':
         fprintf (stderr, "(%d) %s\n", iam, usage_msg);
         return 1;
      case 'D':
         if (parse_to_int (optarg, &dbg_lvl)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%c'\n", iam, optarg, opt);
            return 1;
         }
         break;
      case 'o':
         opt_fname = optarg;
         break;
      default:
         fprintf (stderr, "(%d) internal error: unhandled option '-%c'\n", iam, opt);
         return 1;
      }
   }
   if (optind != argc - 1) {
      fprintf (stderr, "(%d) unexpected number of arguments\n%s\n", iam, usage_msg);
      return 1;
   }
   matrix_fname = argv[optind++];
   return 0;
}

static int grow_per_tracer_opt (int prev_tracer_cnt, int new_tracer_cnt)
{
   char *subname = "grow_per_tracer_opt";

   if (new_tracer_cnt < 1 ||
       (per_tracer_opt = (per_tracer_opt_t *) realloc (per_tracer_opt, (size_t) new_tracer_cnt * sizeof (per_tracer_opt_t))) == NULL) {
      fprintf (stderr, "(%d) realloc failed in %s for grow_per_tracer_opt\n", iam, subname);
      return 1;
   }
   for (int t = prev_tracer_cnt; t < new_tracer_cnt; t++) {
      per_tracer_opt[t].sink_opt = sink_none;
      per_tracer_opt[t].sink_rate = 1.21e-4;   /* radiocarbon decay rate */
      per_tracer_opt[t].sink_depth = 10.0e2;   /* 10 m */
      per_tracer_opt[t].sink_field_name = NULL;
      per_tracer_opt[t].sink_generic_tracer_name = NULL;
      per_tracer_opt[t].sink_generic_tracer_depends_layer_cnt = -1;
      per_tracer_opt[t].pv_field_name = NULL;
      per_tracer_opt[t].d_SF_d_TRACER_field_name = NULL;
   }
   return 0;
}

static int set_opt_defaults (void)
{
   day_cnt = 365.0;
   adv_opt = adv_cent;
   l_adv_enforce_divfree = 1;
   hmix_opt = hmix_isop_file;
   vmix_opt = vmix_file;
   coupled_tracer_cnt = 1;
   if (grow_per_tracer_opt (0, 1)) {
      fprintf (stderr, "(%d) error from grow_per_tracer_opt\n", iam);
      return 1;
   }
   coupled_tracer_opt = coupled_tracer_none;
   return 0;
}

/* keyword tables: option value -> enum */
typedef struct { char *word; int val; } keyword;

static int lookup (const keyword *tab, int n, const char *word, int *val)
{
   for (int e = 0; e < n; e++)
      if (strcmp (tab[e].word, word) == 0) {
         *val = tab[e].val;
         return 0;
      }
   return 1;
}

static int dup_string (char *subname, char *what, char *val, char **dst)
{
   if ((*dst = (char *) malloc (1 + strlen (val))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for %s\n", iam, subname, what);
      return 1;
   }
   strcpy (*dst, val);
   return 0;
}

static int read_opt_file (void)
{
   static const keyword hmix_words[] = { { "none", hmix_none }, { "const", hmix_const }, { "hor_file", hmix_hor_file }, { "isop_file", hmix_isop_file } };
   static const keyword vmix_words[] = { { "none", vmix_none }, { "const", vmix_const }, { "file", vmix_file }, { "matrix_file", vmix_matrix_file } };
   static const keyword sink_words[] = { { "none", sink_none }, { "const", sink_const }, { "const_shallow", sink_const_shallow }, { "file", sink_file }, { "generic_tracer", sink_generic_tracer } };
   static const keyword coupled_words[] = { { "none", coupled_tracer_none }, { "OCMIP_BGC_PO4_DOP", coupled_tracer_OCMIP_BGC_PO4_DOP }, { "DIC_SHADOW_ALK_SHADOW", coupled_tracer_DIC_SHADOW_ALK_SHADOW } };
   char *subname = "read_opt_file";
   FILE *fp;
   char line[MAX_LINE_LEN];
   int line_number = 0;
   int tracer_ind = 0;

   if (opt_fname == NULL)
      return 0;
   if ((fp = fopen (opt_fname, "r")) == NULL) {
      fprintf (stderr, "(%d) fopen failed in %s for %s\n", iam, subname, opt_fname);
      return 1;
   }
   while (fgets (line, MAX_LINE_LEN, fp) != NULL) {
      char *optname, *optval;
      size_t linelen = strlen (line);
      int word;

      line_number++;
      if (linelen == 0)
         continue;
      if (line[linelen - 1] != '\n') {
         fprintf (stderr, "(%d) line number %d in %s too long\n", iam, line_number, opt_fname);
         return 1;
      }
      /* the reference dereferences a NULL optname on a blank line; skip such lines instead */
      if ((optname = strtok (line, " \n")) == NULL)
         continue;
      if ((optval = strtok (NULL, " \n")) == NULL) {
         fprintf (stderr, "(%d) unspecified value for %s\n", iam, optname);
         return 1;
      }
      if (strcmp (optname, "day_cnt") == 0) {
         if (parse_to_double (optval, &day_cnt)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, optval, optname);
            return 1;
         }
      } else if (strcmp (optname, "reg_fname") == 0) {
         if (dup_string (subname, "reg_fname", optval, &reg_fname))
            return 1;
      } else if (strcmp (optname, "circ_fname") == 0) {
         if (dup_string (subname, "circ_fname", optval, &circ_fname))
            return 1;
      } else if (strcmp (optname, "adv_type") == 0) {
         if (strcmp (optval, "none") == 0)
            adv_opt = adv_none;
         else if (strcmp (optval, "donor") == 0)
            adv_opt = adv_donor;
         else if (strncmp (optval, "centered", 4) == 0)      /* "cent", "centred", ... */
            adv_opt = adv_cent;
         else if (strcmp (optval, "upwind3") == 0)
            adv_opt = adv_upwind3;
         else {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
      } else if (strcmp (optname, "l_adv_enforce_divfree") == 0) {
         if (strcmp (optval, "0") == 0)
            l_adv_enforce_divfree = 0;
         else if (strcmp (optval, "1") == 0)
            l_adv_enforce_divfree = 1;
         else {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
      } else if (strcmp (optname, "hmix_type") == 0) {
         if (lookup (hmix_words, 4, optval, &word)) {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
         hmix_opt = (hmix_opt_t) word;
      } else if (strcmp (optname, "vmix_type") == 0) {
         if (lookup (vmix_words, 4, optval, &word)) {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
         vmix_opt = (vmix_opt_t) word;
      } else if (strcmp (optname, "tracer_fname") == 0) {
         if (dup_string (subname, "tracer_fname", optval, &tracer_fname))
            return 1;
      } else if (strcmp (optname, "coupled_tracer_cnt") == 0) {
         int new_coupled_tracer_cnt;

         if (parse_to_int (optval, &new_coupled_tracer_cnt)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, optval, optname);
            return 1;
         }
         /* range first: the reference reallocs to the unchecked count before testing it */
         if ((new_coupled_tracer_cnt < 1) || (new_coupled_tracer_cnt > 2)) {
            fprintf (stderr, "(%d) coupled_tracer_cnt = %d not supported\n", iam, new_coupled_tracer_cnt);
            return 1;
         }
         if (grow_per_tracer_opt (coupled_tracer_cnt, new_coupled_tracer_cnt)) {
            fprintf (stderr, "(%d) error from grow_per_tracer_opt\n", iam);
            return 1;
         }
         coupled_tracer_cnt = new_coupled_tracer_cnt;
      } else if (strcmp (optname, "tracer_ind") == 0) {
         int new_tracer_ind;

         if (parse_to_int (optval, &new_tracer_ind)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, optval, optname);
            return 1;
         }
         if ((new_tracer_ind < 0) || (new_tracer_ind >= coupled_tracer_cnt)) {
            fprintf (stderr, "(%d) tracer_ind = %d out of bounds for coupled_tracer_cnt = %d\n", iam, new_tracer_ind, coupled_tracer_cnt);
            return 1;
         }
         tracer_ind = new_tracer_ind;
      } else if (strcmp (optname, "sink_type") == 0) {
         per_tracer_opt_t *P = &per_tracer_opt[tracer_ind];

         if (lookup (sink_words, 5, optval, &word)) {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
         P->sink_opt = (sink_opt_t) word;
         if ((P->sink_opt == sink_const) || (P->sink_opt == sink_const_shallow)) {
            if ((optval = strtok (NULL, " \n")) == NULL) {
               fprintf (stderr, "(%d) unspecified sink_rate\n", iam);
               return 1;
            }
            if (parse_to_double (optval, &P->sink_rate)) {
               fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, optval, optname);
               return 1;
            }
            if (P->sink_opt == sink_const_shallow) {
               if ((optval = strtok (NULL, " \n")) == NULL) {
                  fprintf (stderr, "(%d) unspecified sink_depth\n", iam);
                  return 1;
               }
               if (parse_to_double (optval, &P->sink_depth)) {
                  fprintf (stderr, "(%d) error parsing argument '%s' for option '%s'\n", iam, optval, optname);
                  return 1;
               }
            }
         }
         if (P->sink_opt == sink_file) {
            if ((optval = strtok (NULL, " \n")) == NULL) {
               fprintf (stderr, "(%d) unspecified sink_field_name\n", iam);
               return 1;
            }
            if (dup_string (subname, "sink_field_name", optval, &P->sink_field_name))
               return 1;
         }
         if (P->sink_opt == sink_generic_tracer) {
            if ((optval = strtok (NULL, " \n")) == NULL) {
               fprintf (stderr, "(%d) unspecified sink_generic_tracer_name\n", iam);
               return 1;
            }
            if (dup_string (subname, "sink_generic_tracer_name", optval, &P->sink_generic_tracer_name))
               return 1;
            if ((optval = strtok (NULL, " \n")) != NULL)
               if (parse_to_int (optval, &P->sink_generic_tracer_depends_layer_cnt)) {
                  fprintf (stderr, "(%d) error parsing sink_generic_tracer_depends_layer_cnt\n", iam);
                  return 1;
               }
         }
      } else if (strcmp (optname, "pv") == 0) {
         if (dup_string (subname, "pv_field_name", optval, &per_tracer_opt[tracer_ind].pv_field_name))
            return 1;
      } else if (strcmp (optname, "sf") == 0) {
         if (dup_string (subname, "d_SF_d_TRACER_field_name", optval, &per_tracer_opt[tracer_ind].d_SF_d_TRACER_field_name))
            return 1;
      } else if (strcmp (optname, "coupled_tracer_type") == 0) {
         if (lookup (coupled_words, 3, optval, &word)) {
            fprintf (stderr, "(%d) unknown %s: %s\n", iam, optname, optval);
            return 1;
         }
         coupled_tracer_opt = (coupled_tracer_opt_t) word;
      } else {
         fprintf (stderr, "(%d) unknown option name: %s\n", iam, optname);
         return 1;
      }
   }
   fclose (fp);

   if (coupled_tracer_cnt == 2)
      if ((coupled_tracer_opt != coupled_tracer_OCMIP_BGC_PO4_DOP) && (coupled_tracer_opt != coupled_tracer_DIC_SHADOW_ALK_SHADOW)) {
         fprintf (stderr, "(%d) coupled_tracer_cnt = 2 only supported for "
                  "coupled_tracer_type = OCMIP_BGC_PO4_DOP, DIC_SHADOW_ALK_SHADOW\n", iam);
         return 1;
      }
   return 0;
}

static void write_opts (void)
{
   static char *adv_names[] = { "none", "donor", "centered", "upwind3" };
   static char *hmix_names[] = { "none", "const", "hor_file", "isop_file" };
   static char *vmix_names[] = { "none", "const", "file", "matrix_file" };
   static char *coupled_names[] = { "none", "OCMIP_BGC_PO4_DOP", "DIC_SHADOW_ALK_SHADOW" };

   if (!dbg_lvl)
      return;
   printf ("(%d) dbg_lvl                    = %d\n", iam, dbg_lvl);
   printf ("(%d) day_cnt                    = %e\n", iam, day_cnt);
   printf ("(%d) reg_fname                  = %s\n", iam, reg_fname ? reg_fname : "none");
   printf ("(%d) circ_fname                 = %s\n", iam, circ_fname);
   printf ("(%d) adv_opt                    = %s\n", iam, adv_names[adv_opt]);
   printf ("(%d) l_adv_enforce_divfree      = %d\n", iam, l_adv_enforce_divfree);
   printf ("(%d) hmix_opt                   = %s\n", iam, hmix_names[hmix_opt]);
   printf ("(%d) vmix_opt                   = %s\n", iam, vmix_names[vmix_opt]);
   printf ("(%d) tracer_fname               = %s\n", iam, tracer_fname ? tracer_fname : "none");
   printf ("(%d) coupled_tracer_cnt         = %d\n", iam, coupled_tracer_cnt);
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      per_tracer_opt_t *P = &per_tracer_opt[t];

      printf ("(%d) options for tracer %d\n", iam, t);
      switch (P->sink_opt) {
      case sink_none:
         printf ("(%d)    sink_opt                = %s\n", iam, "none");
         break;
      case sink_const:
         printf ("(%d)    sink_opt                = %s\n", iam, "const");
         printf ("(%d)    sink_rate               = %e\n", iam, P->sink_rate);
         break;
      case sink_const_shallow:
         printf ("(%d)    sink_opt                = %s\n", iam, "const_shallow");
         printf ("(%d)    sink_rate               = %e\n", iam, P->sink_rate);
         printf ("(%d)    sink_depth              = %e\n", iam, P->sink_depth);
         break;
      case sink_file:
         printf ("(%d)    sink_opt                = %s\n", iam, "file");
         printf ("(%d)    sink_field_name         = %s\n", iam, P->sink_field_name);
         break;
      case sink_generic_tracer:
         printf ("(%d)    sink_opt                = %s\n", iam, "generic_tracer");
         printf ("(%d)    sink_generic_tracer_name= %s\n", iam, P->sink_generic_tracer_name);
         printf ("(%d)    depends_layer_cnt       = %d\n", iam, P->sink_generic_tracer_depends_layer_cnt);
         break;
      }
      printf ("(%d)    pv_field_name           = %s\n", iam, P->pv_field_name ? P->pv_field_name : "none");
      printf ("(%d)    d_SF_d_TRACER_field_name= %s\n", iam, P->d_SF_d_TRACER_field_name ? P->d_SF_d_TRACER_field_name : "none");
   }
   printf ("(%d) coupled_tracer_opt         = %s\n", iam, coupled_names[coupled_tracer_opt]);
   printf ("(%d) matrix_fname               = %s\n\n", iam, matrix_fname);
}

int main (int argc, char *argv[])
{
   iam = 0;
   dbg_lvl = 0;

   if (parse_cmd_line (argc, argv))
      exit (EXIT_FAILURE);
   if (set_opt_defaults ())
      exit (EXIT_FAILURE);
   if (read_opt_file ())
      exit (EXIT_FAILURE);
   /* the reference goes on to nc_open(NULL) when no circulation file was named */
   if (circ_fname == NULL) {
      fprintf (stderr, "(%d) circ_fname not specified (option file line: circ_fname <file>)\n", iam);
      exit (EXIT_FAILURE);
   }
   write_opts ();

   if (get_grid_info (circ_fname, reg_fname))
      exit (EXIT_FAILURE);
   if (put_grid_info (matrix_fname))
      exit (EXIT_FAILURE);
   if (gen_ind_maps ())
      exit (EXIT_FAILURE);
   if (put_ind_maps (matrix_fname))
      exit (EXIT_FAILURE);
   if (gen_sparse_matrix (day_cnt))
      exit (EXIT_FAILURE);
   if (put_sparse_matrix (matrix_fname))
      exit (EXIT_FAILURE);

   free_sparse_matrix ();
   free_ind_maps ();
   free_grid_info ();
   free (per_tracer_opt);
   exit (EXIT_SUCCESS);
}
