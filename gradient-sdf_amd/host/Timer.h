/* tic/toc with the reference's print format "---------- <label>: <ms>ms." (cpp/include/Timer.h:58-81) */
#ifndef GSDF_HOST_TIMER_H_
#define GSDF_HOST_TIMER_H_
#include <chrono>
#include <iostream>
#include <string>
class Timer {
    std::chrono::steady_clock::time_point t0_;
    bool started_ = false;
public:
    void tic() { t0_ = std::chrono::steady_clock::now(); started_ = true; }
    double toc(const std::string& s = "Time elapsed") {
        if (!started_) { std::cout << "Timer was not started, no time could be measured." << std::endl; return 0.; }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
        /* '\n', not std::endl: the frame loop prints three lines per frame, and a flush per line is a system call per line */
        if (el < 1.) std::cout << "---------- " << s << ": " << 1000. * el << "ms.\n";
        else std::cout << "---------- " << s << ": " << el << "s.\n";
        return el;
    }
};
#endif
