/* stub_tsd_hip.h -- read-back interface of the recording stand-in (tests/stub_tsd_hip.c); test infrastructure only */
#ifndef STUB_TSD_HIP_H
#define STUB_TSD_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

enum {
  STUB_CREATE = 0, STUB_DESTROY, STUB_SYNC, STUB_FREE_FOOTPRINT, STUB_PUSH, STUB_RAYCAST, STUB_ICP, STUB_LOCALIZE, STUB_TSDPDF,
  STUB_SET_POSE, STUB_SCAN_STAGE, STUB_SCAN_SUBMIT, STUB_SCAN_COLLECT, STUB_PREREGISTER, STUB_SCAN_BEGIN, STUB_SCAN_FINISH,
  STUB_OP_COUNT
};
#define STUB_LOG_MAX 65536

typedef struct {
  int op;                      /* STUB_* */
  int flag;                    /* STUB_SCAN_SUBMIT: 1 = from the staged scan, 2 = a staged scan was dropped, 0 = plain */
  double tag;                  /* the scan's beam-0 range: the tests number their scans there */
  unsigned long long thread;   /* calling thread */
} stub_entry;

void stub_reset(void);                       /* clear the log and every delay */
void stub_set_delay_us(int op, int us);      /* "device time" of an operation (slept outside the log's lock) */
int stub_log_count(void);
stub_entry stub_log_get(int i);
int stub_live_objects(void);                 /* contexts + sensors created and not destroyed */

#ifdef __cplusplus
}
#endif
#endif
