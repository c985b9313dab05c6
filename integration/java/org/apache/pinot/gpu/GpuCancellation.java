/**
 * Worker thread -> cancel token of the GPU query it is running.  The reference cancels a query by interrupting its worker threads
 * (the query killer / timeouts: BaseCombineOperator.java:147-156 cancels the futures; operators notice through
 * Thread.interrupted() in BaseOperator#nextBlock, BaseOperator.java:44-46, and throw EarlyTerminationException).  A thread blocked
 * inside PinotGpu.queryExec does not poll its interrupt flag, so whoever interrupts it also sets its token:
 *
 *   GpuCancellation.cancel(thread)     - called next to Future#cancel(true) / Thread#interrupt for the workers of a query
 *   PinotGpu.queryExec(..., token)     - polls the token between launches and throws EarlyTerminationException (PG_ERR_CANCELLED)
 *
 * Tokens are owned by GpuGroupByOperator#getNextBlock (create -> register -> exec -> unregister -> destroy); cancel() only touches a
 * token while it is registered, under the same lock that unregisters it, so a token is never used after its destruction.
 */
package org.apache.pinot.gpu;

import java.util.HashMap;
import java.util.Map;

public final class GpuCancellation {
  private static final Map<Thread, Long> TOKENS = new HashMap<>();

  private GpuCancellation() {
  }

  public static void register(Thread worker, long token) {
    synchronized (TOKENS) {
      TOKENS.put(worker, token);
    }
  }

  public static void unregister(Thread worker) {
    synchronized (TOKENS) {
      TOKENS.remove(worker);
    }
  }

  /** Requests cancellation of the GPU query `worker` is executing, if any.  Returns whether a token was set. */
  public static boolean cancel(Thread worker) {
    synchronized (TOKENS) {
      Long token = TOKENS.get(worker);
      if (token == null) {
        return false;
      }
      PinotGpu.cancelRequest(token);
      return true;
    }
  }

  /** Cancels every GPU query in flight (server shutdown). */
  public static int cancelAll() {
    synchronized (TOKENS) {
      for (long token : TOKENS.values()) {
        PinotGpu.cancelRequest(token);
      }
      return TOKENS.size();
    }
  }
}
