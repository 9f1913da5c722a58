/* Stand-in of every build of the reference's program in this image (the drop-in program and the all-reference builds).
 *
 * Backend of netcdf_rec.f90: the NetCDF library is absent from this image, so the reference's output modules
 * (src/modstat_nc.f90, the one nf90_open in src/initfac.f90:267) link against a stand-in that keeps the dimension / variable
 * tables of each file in memory and APPENDS what nf90_put_var is handed to the file itself as a flat record stream:
 *
 *   "UDNC" | int32 kind | ...
 *     kind 1  dimension   int32 id, int32 len, int32 namelen, name
 *     kind 2  variable    int32 id, int32 xtype, int32 ndims, int32 dimids[ndims], int32 namelen, name
 *     kind 3  data        int32 varid, int32 nstart, int32 start[nstart], int32 rank, int32 shape[rank], float64 values[prod(shape)]
 *
 * tests/refdump.py (read_ncrec) reads it back.  Values are kept in float64, i.e. BEFORE NetCDF's conversion to NF90_FLOAT: two
 * builds of the same driver can be compared to round-off.  No arithmetic, no uDALES code. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXF 64
#define MAXD 32
#define MAXV 512
#define NAMEL 96

#define MAXT 8192
typedef struct {
    FILE *f;
    char path[1024];
    int ndim, nvar, unlimited, nrec;
    char dname[MAXD][NAMEL];
    int dlen[MAXD];
    char vname[MAXV][NAMEL];
    int vnd[MAXV], vlast[MAXV];      /* number of dimensions, id of the slowest one */
    int timevar, ntime;              /* the 1-D variable over the record dimension and what was written to it: what a re-open reads back */
    double timeval[MAXT];
} ncfile;

static ncfile files[MAXF];

static ncfile *get(int ncid) { return (ncid >= 1 && ncid <= MAXF && files[ncid - 1].f) ? &files[ncid - 1] : NULL; }

static void trimcpy(char *dst, const char *src, int n)
{
    while (n > 0 && (src[n - 1] == ' ' || src[n - 1] == '\0')) n--;
    if (n >= NAMEL) n = NAMEL - 1;
    memcpy(dst, src, (size_t)n);
    dst[n] = 0;
}

static void w32(FILE *f, int32_t v) { fwrite(&v, 4, 1, f); }

static void head(FILE *f, int kind)
{
    fwrite("UDNC", 1, 4, f);
    w32(f, kind);
}

int udnc_create(const char *path, int n)
{
    char p[1024];
    int id;
    if (n > 1023) n = 1023;
    memcpy(p, path, (size_t)n);
    while (n > 0 && p[n - 1] == ' ') n--;
    p[n] = 0;
    for (id = 0; id < MAXF; id++)
        if (!files[id].f) break;
    if (id == MAXF) return -1;
    memset(&files[id], 0, sizeof(ncfile));
    files[id].f = fopen(p, "wb");
    files[id].unlimited = -1;
    strcpy(files[id].path, p);
    return files[id].f ? id + 1 : -1;
}

/* nf90_open of a file this process created and still holds (src/modfielddump.f90:313-321 opens its file twice): the same tables */
int udnc_reopen(const char *path, int n)
{
    char p[1024];
    if (n > 1023) n = 1023;
    memcpy(p, path, (size_t)n);
    while (n > 0 && p[n - 1] == ' ') n--;
    p[n] = 0;
    for (int id = 0; id < MAXF; id++)
        if (files[id].f && !strcmp(files[id].path, p)) return id + 1;
    return -1;
}

int udnc_time_values(int ncid, int varid, int n, double *out)
{
    ncfile *c = get(ncid);
    if (!c || varid != c->timevar) return -1;
    for (int i = 0; i < n; i++) out[i] = i < c->ntime ? c->timeval[i] : 0.;
    return 0;
}

int udnc_def_dim(int ncid, const char *name, int n, int len)
{
    ncfile *c = get(ncid);
    if (!c) return -1;
    int id = c->ndim;
    if (id >= MAXD) return -1;
    trimcpy(c->dname[id], name, n);
    c->dlen[id] = len;
    if (len == 0) c->unlimited = id + 1;
    c->ndim++;
    head(c->f, 1);
    w32(c->f, id + 1); w32(c->f, len); w32(c->f, (int32_t)strlen(c->dname[id]));
    fwrite(c->dname[id], 1, strlen(c->dname[id]), c->f);
    return id + 1;
}

int udnc_inq_dimid(int ncid, const char *name, int n)
{
    ncfile *c = get(ncid);
    char t[NAMEL];
    if (!c) return -1;
    trimcpy(t, name, n);
    for (int i = 0; i < c->ndim; i++)
        if (!strcmp(t, c->dname[i])) return i + 1;
    return -1;
}

int udnc_dim_len(int ncid, int dimid)
{
    ncfile *c = get(ncid);
    if (!c || dimid < 1 || dimid > c->ndim) return -1;
    return dimid == c->unlimited ? c->nrec : c->dlen[dimid - 1];
}

int udnc_unlimited(int ncid) { return get(ncid) ? get(ncid)->unlimited : -1; }

int udnc_def_var(int ncid, const char *name, int n, int xtype, int ndims, const int *dimids)
{
    ncfile *c = get(ncid);
    if (!c) return -1;
    int id = c->nvar;
    if (id >= MAXV) return -1;
    trimcpy(c->vname[id], name, n);
    c->vnd[id] = ndims;
    c->vlast[id] = ndims > 0 ? dimids[ndims - 1] : -1;
    if (ndims == 1 && dimids[0] == c->unlimited && !c->timevar) c->timevar = id + 1;
    c->nvar++;
    head(c->f, 2);
    w32(c->f, id + 1); w32(c->f, xtype); w32(c->f, ndims);
    for (int i = 0; i < ndims; i++) w32(c->f, dimids[i]);
    w32(c->f, (int32_t)strlen(c->vname[id]));
    fwrite(c->vname[id], 1, strlen(c->vname[id]), c->f);
    return id + 1;
}

int udnc_inq_varid(int ncid, const char *name, int n)
{
    ncfile *c = get(ncid);
    char t[NAMEL];
    if (!c) return -1;
    trimcpy(t, name, n);
    for (int i = 0; i < c->nvar; i++)
        if (!strcmp(t, c->vname[i])) return i + 1;
    return -1;
}

int udnc_put(int ncid, int varid, int nstart, const int *start, int rank, const int *shape, const double *v)
{
    ncfile *c = get(ncid);
    size_t cnt = 1;
    if (!c || varid < 1 || varid > c->nvar) return -1;
    if (c->vlast[varid - 1] == c->unlimited && nstart == c->vnd[varid - 1] && nstart > 0) {      /* a record variable: the record number */
        int rec = start[nstart - 1];
        if (rec > c->nrec) c->nrec = rec;
        if (varid == c->timevar && rec >= 1 && rec <= MAXT) { c->timeval[rec - 1] = v[0]; if (rec > c->ntime) c->ntime = rec; }
    }
    head(c->f, 3);
    w32(c->f, varid); w32(c->f, nstart);
    for (int i = 0; i < nstart; i++) w32(c->f, start[i]);
    w32(c->f, rank);
    for (int i = 0; i < rank; i++) { w32(c->f, shape[i]); cnt *= (size_t)shape[i]; }
    fwrite(v, 8, cnt, c->f);
    return 0;
}

int udnc_sync(int ncid)
{
    if (ncid < 1 || ncid > MAXF || !files[ncid - 1].f) return -1;
    fflush(files[ncid - 1].f);
    return 0;
}

int udnc_close(int ncid)
{
    if (ncid < 1 || ncid > MAXF) return -1;
    if (!files[ncid - 1].f) return 0;      /* (a file opened twice has one handle here) */
    fclose(files[ncid - 1].f);
    files[ncid - 1].f = NULL;
    return 0;
}
