/* Strict numeric argument parsing with the reference's semantics (reference
 * src/misc.c:11-95): base-0 strtol / strtod, reject empty input, trailing characters and
 * out-of-range values; 0 = ok, 1 = error with an "(iam) name:..." message on stderr. */
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

#include "nkp_host.h"

int dbg_lvl = 0;
int iam = 0;

typedef enum { AS_LONG, AS_DOUBLE } numkind_t;

static int parse_number (const char *who, char *str, numkind_t kind, long *lval, double *dval)
{
   char *end = NULL;

   if (str == NULL || str[0] == '\0') {
      fprintf (stderr, "(%d) %s:nothing to parse\n", iam, who);
      return 1;
   }
   errno = 0;
   if (kind == AS_LONG)
      *lval = strtol (str, &end, 0);
   else
      *dval = strtod (str, &end);
   if (errno == ERANGE) {
      fprintf (stderr, "(%d) %s:ERANGE error parsing '%s'\n", iam, who, str);
      return 1;
   }
   if (*end != '\0') {
      fprintf (stderr, "(%d) %s:unexpected character '%c' parsing '%s'\n", iam, who, *end, str);
      return 1;
   }
   return 0;
}

int parse_to_long (char *str, long *val) { return parse_number ("parse_to_long", str, AS_LONG, val, NULL); }

int parse_to_double (char *str, double *val) { return parse_number ("parse_to_double", str, AS_DOUBLE, NULL, val); }

int parse_to_int (char *str, int *val)
{
   const char *who = "parse_to_int";
   long wide;

   if (str == NULL || str[0] == '\0') {
      fprintf (stderr, "(%d) %s:nothing to parse\n", iam, who);
      return 1;
   }
   if (parse_to_long (str, &wide)) {
      fprintf (stderr, "(%d) %s:error from parse_to_long\n", iam, who);
      return 1;
   }
   if (wide < INT_MIN || wide > INT_MAX) {
      fprintf (stderr, "(%d) %s:value %ld out of int range\n", iam, who, wide);
      return 1;
   }
   *val = (int) wide;
   return 0;
}
